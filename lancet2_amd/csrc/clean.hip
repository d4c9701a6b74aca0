// Graph cleaning + haplotype enumeration on gfx950: the per-window serial heart of cbdg --
// MarkConnectedComponents, FindSource/FindSink, PruneComponent (CompressGraph / RemoveLowCovNodes /
// RemoveTips), BuildTraversalIndex, HasCycle, ComputeGraphComplexity, MaxFlow::NextPath, BuildHaplotypes
// (cbdg/graph.cpp:142-235, :392-924; traversal_index.cpp; cycle_finder.cpp; graph_complexity.cpp;
// max_flow.cpp; path.cpp).
//
// One wavefront per window attempt.  The algorithm is order dependent by construction (floor-averaged
// coverage merges, edge-list order, first-found walks), so its control flow runs in the CANONICAL order of
// DESIGN.md, executed UNIFORMLY by all 64 lanes (same addresses, same values: a wave instruction costs the
// same with one lane or 64), and every loop over all nodes / reference k-mers / bases -- flag scans,
// initialisation, candidate collection in index order, byte copies and compares -- is split across the
// lanes with ballots keeping the index order.  It works on the compact arrays build.hip produced; node sequences are never
// materialised during cleaning -- a node is a doubly linked list of SLICES of original k-mers
// (start, len, revcomp), which reproduces Kmer::Merge's string semantics (kmer.cpp:48-109) exactly,
// including for the reference's stored-sign edge quirk.  Haplotype bases are spelled from the batch's
// ref/read bytes only when a walk is emitted.
//
// Three places where the serial order is kept but the work is not serial (each is re-derived from memory or proven
// in place, so a stale hint or an unusual graph only costs speed):
//   * CompressNode: chains are followed through a next-state table in LDS, 64 positions at a time, every lane
//     validates its position's merge against the records in HBM, and only the floor-rounded running averages run as
//     a (scalar) recurrence (compress_walk_par); a lane-parallel exact negative filter skips nodes that cannot merge;
//   * MaxFlow::NextPath: queue entries of one level that stand on the same state are folded into the earliest one (two,
//     if the earliest has not crossed a new edge yet), with multiplicities standing in for the reference's visit count;
//   * slice lists are ranked by pointer jumping in LDS before haplotypes are spelled.
// Loops over the nodes of a component (traversal index, complexity metrics, confidence tables) run one lane per node;
// a loop that walks global memory serially costs a memory round trip per iteration and a kernel of one wavefront per
// window lasts as long as its slowest window.
#include <algorithm>
#include <vector>

#include "graph_ws.h"

namespace ma {

namespace {

#ifdef MA_PROFILE
__device__ unsigned long long g_cprof[16];
__device__ unsigned long long g_cwin[4096 * 4];
__device__ unsigned g_cmerge[4096 * 8];
__device__ unsigned long long g_ctime[4096 * 6];
#define TSTAMP(v) unsigned long long v = __builtin_amdgcn_s_memtime()
#define TACC(slot, a, b) g.dbg_t[slot] += (b) - (a)
#define CCOUNT(slot) do {} while (0)
#else
#define CCOUNT(slot) do {} while (0)
#define TSTAMP(v) do {} while (0)
#define TACC(slot, a, b) do {} while (0)
#endif
constexpr int kMaxWalks = 64;
constexpr u32 kNoParent = 0xFFFFFFFFu;

struct Win {
  // inputs
  const u8* refb;   // window reference bytes
  const u8* readb;  // first read byte of the window
  u32 ref_len;
  int k, S;
  u32 n;            // number of nodes
  u32 nc;
  u32 min_node_cov, min_anchor_cov;
  // node arrays
  u32* cnt;
  u32* role;
  const u32* src;
  u8* label;
  const u8* sign;
  u8* nedge;
  u32* edge;
  u32* comp;
  u32* len;
  u8* alive;
  u32* head;
  u32* tail;
  u32* snext;
  u32* sprev;
  u32* sdesc;         // slice = (start, len, revcomp) of a BASE STRING: sd_make()
  const u32* bsrc;    // base string of slice s: bit31 = read buffer (else ref), bit30 = the window's string pool; low bits = byte offset
  const u32* blen;    // its length; null: every base string is an original k-mer (length k)
  const u8* bsign;    // 1: the bytes are the stored orientation, 0: its reverse complement (a k-mer first seen on the other strand)
  const u8* pool;     // merged strings written by k_clean_chains (null on the raw graph)
  u32 ecap;           // edge slots per node in `edge`
  bool lds;           // every array the lanes cooperate through lives in LDS
  u32 dbg_why;        // (diagnostics) source line of the capacity that made the compact route give the window up
  u32 ek;             // scratch layout: the four per-edge arrays hold ek * nc entries each, the walk pool wk * nc
  u32 wk;
  u32* pieces;        // (LDS kernels) piece table of the walk being spelled: [0] = count, then two words per piece
  u32 piece_cap;
  u32 link_cap;       // u32 words behind `link`
  uint4* arena;       // search arena of MaxFlow::NextPath
  u32 ac;             // its capacity (records)
  u32* scratch;
  u32* link;  // LDS, 8 KB: component labels, then the chain-following table, then the slice ranks
  bool ranked;  // link holds rank_slices() of the current graph
  i64 source, sink;
  u32 flags;  // bit2: capacity overflow
#ifdef MA_PROFILE
  u32 dbg_phase, dbg_merges[2], dbg_maxwalk[2], dbg_walks[2];
  unsigned long long dbg_t[6];
  unsigned long long dbg_ph[16];
#endif
};

// slice descriptor: start (15 bit) | length (15 bit) << 15 | reverse complement << 30
__device__ __forceinline__ u32 sd_make(u32 st, u32 ln, u32 rc) { return st | (ln << 15) | (rc << 30); }
__device__ __forceinline__ u32 sd_st(u32 d) { return d & 0x7FFFu; }
__device__ __forceinline__ u32 sd_ln(u32 d) { return (d >> 15) & 0x7FFFu; }
__device__ __forceinline__ u32 sd_rc(u32 d) { return (d >> 30) & 1u; }
__device__ __forceinline__ u32 base_len(const Win& g, u32 s) { return g.blen ? g.blen[s] : static_cast<u32>(g.k); }

__device__ __forceinline__ u32 kind_rev(u32 kind) { return (((kind & 1u) ^ 1u) << 1) | (((kind >> 1) & 1u) ^ 1u); }
__device__ __forceinline__ u32 mirror_of(u32 self, u32 val) { return (self << 2) | kind_rev(val & 3u); }

__device__ __forceinline__ u32 nd_total(const Win& g, u32 i) {
  u32 t = 0;
  for (int s = 0; s < g.S; ++s) t += g.cnt[i * g.S + s];
  return t;
}
__device__ __forceinline__ bool nd_all_singletons(const Win& g, u32 i) {
  bool any = false, all = true;
  for (int s = 0; s < g.S; ++s) {
    u32 const c = g.cnt[i * g.S + s];
    any |= c > 0;
    all &= c <= 1;
  }
  return any && all;
}
// Node::Confidence (node.cpp:59-79)
__device__ __forceinline__ u32 nd_confidence(const Win& g, u32 i) {
  if (nd_all_singletons(g, i)) return 1;
  u32 const total = nd_total(g, i);
  if (total == 0) return 0;
  int confirming = 0;
  for (int s = 0; s < g.S; ++s) confirming += g.cnt[i * g.S + s] > 0;
  f64 const denom = static_cast<f64>(g.S > 1 ? g.S : 1);
  f64 const concordance = static_cast<f64>(confirming) / denom;
  u32 const bonus = (g.label[i] & 1u) ? 1u : 0u;
  return static_cast<u32>(static_cast<f64>(total) * concordance) + bonus;
}

__device__ __forceinline__ void emplace_edge(Win& g, u32 i, u32 val) {  // node.h:59-64
  u32* e = g.edge + i * g.ecap;
  int const n = g.nedge[i];
  for (int x = 0; x < n; ++x)
    if (e[x] == val) return;
  if (n >= static_cast<int>(g.ecap)) {
    g.flags |= 4u;
    g.dbg_why = __LINE__;
    return;
  }
  e[n] = val;
  g.nedge[i] = static_cast<u8>(n + 1);
}
__device__ __forceinline__ void erase_edge(Win& g, u32 i, u32 val) {  // node.h:66-71
  u32* e = g.edge + i * g.ecap;
  int const n = g.nedge[i];
  for (int x = 0; x < n; ++x)
    if (e[x] == val) {
      for (int y = x; y + 1 < n; ++y) e[y] = e[y + 1];
      g.nedge[i] = static_cast<u8>(n - 1);
      return;
    }
}
__device__ __forceinline__ bool has_self_loop(const Win& g, u32 i) {
  const u32* e = g.edge + i * g.ecap;
  for (int x = 0; x < g.nedge[i]; ++x)
    if ((e[x] >> 2) == i) return true;
  return false;
}
// FindEdgesInDirection (node.cpp:118-127): count + first match
__device__ __forceinline__ int edges_in_dir(const Win& g, u32 i, bool dflt, u32* first) {
  u32 const exp_minus = dflt ? (g.sign[i] ? 0u : 1u) : (g.sign[i] ? 1u : 0u);
  const u32* e = g.edge + i * g.ecap;
  int c = 0;
  for (int x = 0; x < g.nedge[i]; ++x)
    if (((e[x] >> 1) & 1u) == exp_minus) {
      if (c == 0) *first = e[x];
      c++;
    }
  return c;
}

// Graph::RemoveNode (graph.cpp:347-361)
__device__ __forceinline__ void remove_node(Win& g, u32 i) {
  if (!g.alive[i]) return;
  const u32* e = g.edge + i * g.ecap;
  for (int x = 0; x < g.nedge[i]; ++x) {
    u32 const d = e[x] >> 2;
    if (d == i) continue;
    if (g.alive[d]) erase_edge(g, d, mirror_of(i, e[x]));
  }
  g.alive[i] = 0;
  g.nedge[i] = 0;
}

// lanes of the wave cooperate on index ranges; stores of one lane are made visible to the others here
// lanes of the wave cooperate through memory: a lane's stores must have landed before another lane's loads.  On the LDS
// image that only means "LDS operations issued so far are done" (the LDS unit serves a wave in order anyway); with work
// lists in HBM it is a fence -- which also waits for every outstanding store to HBM to be acknowledged (~2 us a time:
// haplotype bases, run records), so the LDS kernels must not use it.
__device__ __forceinline__ void wave_sync_mem(bool lds_only) {
  if (lds_only) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else __threadfence_block();
}
__device__ __forceinline__ u32 lane_id() { return threadIdx.x; }
// append the indices i (in increasing order) whose flag is set to list[]: 64 indices per ballot
__device__ __forceinline__ void ordered_append(bool flag, u32 i, u32* list, u32& count) {
  unsigned long long const m = __ballot(flag);
  if (flag) list[count + __popcll(m & ((1ull << lane_id()) - 1ull))] = i;
  count += static_cast<u32>(__popcll(m));
}

// RemoveLowCovNodes (graph.cpp:363-390)
__device__ __forceinline__ void remove_low_cov(Win& g, u32 comp) {
  u32* rm = g.scratch;
  u32 nrm = 0;
  for (u32 base = 0; base < g.n; base += 64) {
    u32 const i = base + lane_id();
    bool flag = false;
    if (i < g.n && g.alive[i] && g.comp[i] == comp && static_cast<i64>(i) != g.source && static_cast<i64>(i) != g.sink)
      flag = nd_all_singletons(g, i) || nd_total(g, i) < g.min_node_cov;
    ordered_append(flag, i, rm, nrm);
  }
  wave_sync_mem(g.lds);
  for (u32 x = 0; x < nrm; ++x) remove_node(g, rm[x]);
}

// ---- slice lists: a node's sequence in its stored orientation ----
__device__ __forceinline__ void slices_reverse(Win& g, u32 node) {  // sequence := RevComp(sequence)
  u32 s = g.head[node];
  while (s != kNoNode) {
    u32 const nx = g.snext[s];
    g.snext[s] = g.sprev[s];
    g.sprev[s] = nx;
    u32 const d = g.sdesc[s];
    u32 const st = sd_st(d), ln = sd_ln(d), rc = sd_rc(d);
    g.sdesc[s] = sd_make(base_len(g, s) - st - ln, ln, rc ^ 1u);
    s = nx;
  }
  u32 const h = g.head[node];
  g.head[node] = g.tail[node];
  g.tail[node] = h;
}
__device__ __forceinline__ void slices_drop_front(Win& g, u32 node, u32 nb) {
  while (nb > 0) {
    u32 const s = g.head[node];
    u32 const d = g.sdesc[s];
    u32 const st = sd_st(d), ln = sd_ln(d);
    if (ln <= nb) {
      nb -= ln;
      u32 const nx = g.snext[s];
      g.head[node] = nx;
      if (nx != kNoNode) g.sprev[nx] = kNoNode; else g.tail[node] = kNoNode;
    } else {
      g.sdesc[s] = sd_make(st + nb, ln - nb, sd_rc(d));
      nb = 0;
    }
  }
}
__device__ __forceinline__ void slices_drop_back(Win& g, u32 node, u32 nb) {
  while (nb > 0) {
    u32 const s = g.tail[node];
    u32 const d = g.sdesc[s];
    u32 const st = sd_st(d), ln = sd_ln(d);
    if (ln <= nb) {
      nb -= ln;
      u32 const pv = g.sprev[s];
      g.tail[node] = pv;
      if (pv != kNoNode) g.snext[pv] = kNoNode; else g.head[node] = kNoNode;
    } else {
      g.sdesc[s] = sd_make(st, ln - nb, sd_rc(d));
      nb = 0;
    }
  }
}

// floor((a * la + b * lb) / (la + lb)) of node.cpp:93-104; the 64-bit division is ~10x the cost of the 32-bit one
__device__ __forceinline__ u32 weighted_floor_avg(u32 a, u64 la, u32 b, u64 lb) {
  u64 const num = static_cast<u64>(a) * la + static_cast<u64>(b) * lb, den = la + lb;
  if ((num >> 32) == 0 && (den >> 32) == 0) return static_cast<u32>(num) / static_cast<u32>(den);
  return static_cast<u32>(num / den);
}

// Node::Merge (node.cpp:81-112) + Kmer::Merge / MergeCords (kmer.cpp:48-109)
__device__ __forceinline__ void merge_node(Win& g, u32 x, u32 b, u32 kind) {
  u32 const K1 = static_cast<u32>(g.k) - 1;
  u32 const blen = g.len[b];
  bool const append = kind == 0 || kind == 1;  // PLUS_PLUS / PLUS_MINUS append, MINUS_* prepend
  bool const rc = kind == 1 || kind == 2;      // PLUS_MINUS / MINUS_PLUS use RevComp(other)
  if (rc) slices_reverse(g, b);
  if (append) {
    slices_drop_front(g, b, K1);  // NonOvlSuffix
    if (g.head[b] != kNoNode) {
      g.snext[g.tail[x]] = g.head[b];
      g.sprev[g.head[b]] = g.tail[x];
      g.tail[x] = g.tail[b];
    }
  } else {
    slices_drop_back(g, b, K1);  // NonOvlPrefix
    if (g.head[b] != kNoNode) {
      g.sprev[g.head[x]] = g.tail[b];
      g.snext[g.tail[b]] = g.head[x];
      g.head[x] = g.head[b];
    }
  }
  g.head[b] = g.tail[b] = kNoNode;
  g.len[x] += blen - K1;
  g.label[x] |= g.label[b];
  u64 const this_len = g.len[x];  // length AFTER the merge (node.cpp:91)
  u64 const other_len = blen;
  for (int s = 0; s < g.S; ++s)
    g.cnt[x * g.S + s] = weighted_floor_avg(g.cnt[x * g.S + s], this_len, g.cnt[b * g.S + s], other_len);
  for (int r = 0; r < 2; ++r)
    g.role[x * 2 + r] = weighted_floor_avg(g.role[x * 2 + r], this_len, g.role[b * 2 + r], other_len);
}

// IsPotentialBuddyEdge (graph.cpp:758-799); conn = edge value stored at src
__device__ __forceinline__ bool is_potential_buddy(const Win& g, u32 src, u32 conn) {
  u32 const nb = conn >> 2;
  if (g.nedge[src] == 1 && g.nedge[nb] == 1) {
    if ((g.edge[src * g.ecap] >> 2) == nb && (g.edge[nb * g.ecap] >> 2) == src) return false;
  }
  if (g.nedge[nb] > 2 || g.nedge[nb] == 0 || has_self_loop(g, nb)) return false;
  u32 const expected = mirror_of(src, conn);  // nbour -> src
  // direction of nbour whose SrcSign equals the mirror's SrcSign
  u32 const exp_src_minus = (expected >> 1) & 1u;
  bool const dir_dflt = (exp_src_minus == 0u) == (g.sign[nb] != 0);
  u32 f = 0;
  int const c1 = edges_in_dir(g, nb, dir_dflt, &f);
  if (c1 != 1 || f != expected) return false;
  u32 f2 = 0;
  int const c2 = edges_in_dir(g, nb, !dir_dflt, &f2);
  if (c2 != 1 || (f2 >> 2) == src) return false;
  return g.nedge[f2 >> 2] <= 2;
}

// FindCompressibleEdge (graph.cpp:688-717)
__device__ __forceinline__ bool find_compressible_edge(const Win& g, u32 src, bool dflt, u32* out) {
  if (g.nedge[src] > 2 || g.nedge[src] == 0 || has_self_loop(g, src)) return false;
  if (static_cast<i64>(src) == g.source || static_cast<i64>(src) == g.sink) return false;
  u32 cand = 0;
  if (edges_in_dir(g, src, dflt, &cand) != 1) return false;
  u32 const d = cand >> 2;
  if (static_cast<i64>(d) == g.source || static_cast<i64>(d) == g.sink) return false;
  if (!is_potential_buddy(g, src, cand)) return false;
  u32 opp = 0;
  int const co = edges_in_dir(g, src, !dflt, &opp);
  if (co == 0) {
    *out = cand;
    return true;
  }
  if (co > 1) return false;
  if (!is_potential_buddy(g, src, opp)) return false;
  *out = cand;
  return true;
}

// CompressNode (graph.cpp:600-645)
__device__ __forceinline__ void compress_node(Win& g, u32 nid, bool dflt, u8* absorbed) {
  u32 s2o = 0;
  while (find_compressible_edge(g, nid, dflt, &s2o)) {
    u32 const ob = s2o >> 2, kind = s2o & 3u;
    merge_node(g, nid, ob, kind);
    erase_edge(g, nid, s2o);
    u32 const src_minus = (kind >> 1) & 1u, dst_minus = kind & 1u;
    u32 const mirror = mirror_of(nid, s2o);
    int const nob = g.nedge[ob];
    for (int x = 0; x < nob; ++x) {
      u32 const o2n = g.edge[ob * g.ecap + x];
      if (o2n == mirror) continue;
      u32 const nbd = o2n >> 2;
      u32 const o2n_src_minus = (o2n >> 1) & 1u, o2n_dst_minus = o2n & 1u;
      u32 const ne_src_minus = (dst_minus != o2n_src_minus) ? (src_minus ^ 1u) : src_minus;
      u32 const s2n = (nbd << 2) | (ne_src_minus << 1) | o2n_dst_minus;
      emplace_edge(g, nid, s2n);
      emplace_edge(g, nbd, mirror_of(nid, s2n));
      erase_edge(g, nbd, mirror_of(ob, o2n));
    }
    CCOUNT(12);
    absorbed[ob] = 1;
  }
}

// is_potential_buddy with the far neighbour reported (same tests, same order)
__device__ __forceinline__ bool is_potential_buddy_f2(const Win& g, u32 src, u32 conn, u32* f2node) {
  u32 const nb = conn >> 2;
  *f2node = kNoNode;
  if (g.nedge[src] == 1 && g.nedge[nb] == 1) {
    if ((g.edge[src * g.ecap] >> 2) == nb && (g.edge[nb * g.ecap] >> 2) == src) return false;
  }
  if (g.nedge[nb] > 2 || g.nedge[nb] == 0 || has_self_loop(g, nb)) return false;
  u32 const expected = mirror_of(src, conn);
  u32 const exp_src_minus = (expected >> 1) & 1u;
  bool const dir_dflt = (exp_src_minus == 0u) == (g.sign[nb] != 0);
  u32 f = 0;
  int const c1 = edges_in_dir(g, nb, dir_dflt, &f);
  if (c1 != 1 || f != expected) return false;
  u32 f2 = 0;
  int const c2 = edges_in_dir(g, nb, !dir_dflt, &f2);
  if (c2 != 1 || (f2 >> 2) == src) return false;
  *f2node = f2 >> 2;
  return g.nedge[f2 >> 2] <= 2;
}

// CompressNode for the regular case -- a walk down a chain of degree-2 nodes -- with the absorbing
// node and the next candidate held in registers: one batch of independent loads per merge instead of
// ~60 dependent ones.  It only ever takes POSITIVE decisions (the exact find_compressible_edge /
// is_potential_buddy predicates evaluated on register copies of what memory holds); as soon as a test
// fails or anything unusual shows up (node coincidences of a cycle, duplicate edges) it writes its
// state back and hands over to the generic compress_node, which re-evaluates from memory.
template <int SMAX>  // per-sample registers of the walk: 2 covers tumour/normal, kMaxSamples everything else
__device__ __forceinline__ void compress_node_fast(Win& g, u32 nid, bool dflt, u8* absorbed) {
  // NOTE: no dynamically indexed local arrays in here -- they would live in scratch (HBM latency)
  if (static_cast<i64>(nid) == g.source || static_cast<i64>(nid) == g.sink) return;
  u32 xn = g.nedge[nid];
  if (xn > 2 || xn == 0) return;
  u32 const K1 = static_cast<u32>(g.k) - 1;
  int const S = g.S;
  u32 xe0, xe1;
  {
    uint2 const ev = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(nid) * g.ecap);
    xe0 = ev.x;
    xe1 = ev.y;
  }
  u32 const xsign = g.sign[nid];
  u32 const exp_minus = dflt ? (xsign ? 0u : 1u) : (xsign ? 1u : 0u);
#ifdef MA_PROFILE
  u32 dbg_nm = 0;
#endif
  bool dirty = false;
  // The whole record of nid is fetched here, with its edges, and every load of one walk step is issued in two
  // unconditional straight-line batches (the record of the neighbour b; then the far neighbour + b's slice): a
  // load behind a branch cannot be hoisted, and the compiler then waits for memory at every use.
  bool const xloaded = true;
  u32 xlen = g.len[nid], xlabel = g.label[nid], xhead = g.head[nid], xtail = g.tail[nid];
  u32 xcnt[SMAX], xrole0 = g.role[nid * 2], xrole1 = g.role[nid * 2 + 1];
#pragma unroll
  for (int t = 0; t < SMAX; ++t) {
    u32 const v = g.cnt[nid * S + (t < S ? t : 0)];
    xcnt[t] = t < S ? v : 0u;
  }
  // opposite side: buddy test cached per edge value (its inputs do not change while we walk the other way)
  bool opp_known = false, opp_ok = false;
  u32 opp_val = 0, opp_nb = kNoNode, opp_f2 = kNoNode;
  auto flush = [&]() {
    if (!dirty) return;
    g.nedge[nid] = static_cast<u8>(xn);
    g.edge[static_cast<size_t>(nid) * g.ecap] = xe0;
    g.edge[static_cast<size_t>(nid) * g.ecap + 1] = xe1;
    if (xloaded) {
      g.len[nid] = xlen;
      g.label[nid] = static_cast<u8>(xlabel);
#pragma unroll
      for (int t = 0; t < SMAX; ++t)
        if (t < S) g.cnt[nid * S + t] = xcnt[t];
      g.role[nid * 2] = xrole0;
      g.role[nid * 2 + 1] = xrole1;
      g.head[nid] = xhead;
      g.tail[nid] = xtail;
    }
    dirty = false;
  };
  while (true) {
    // ---- find_compressible_edge(nid, dflt) on the register copy ----
    if (xn > 2 || xn == 0) break;
    if ((xe0 >> 2) == nid || (xn == 2 && (xe1 >> 2) == nid)) break;
    bool const d0 = ((xe0 >> 1) & 1u) == exp_minus;
    bool const d1 = xn == 2 && ((xe1 >> 1) & 1u) == exp_minus;
    int const cdir = (d0 ? 1 : 0) + (d1 ? 1 : 0);
    int const copp = static_cast<int>(xn) - cdir;
    if (cdir != 1) break;
    u32 const cand = d0 ? xe0 : xe1;
    u32 const opp = d0 ? xe1 : xe0;  // only meaningful when copp == 1
    u32 const d = cand >> 2;
    if (static_cast<i64>(d) == g.source || static_cast<i64>(d) == g.sink) break;
    // ---- batch 1: the record of b, straight from memory (the previous step's stores to it are ordered before) ----
    u32 const bn = g.nedge[d];
    uint2 const bev = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(d) * g.ecap);
    u32 const be0 = bev.x, be1 = bev.y;
    u32 const bsign = g.sign[d];
    u32 const blen = g.len[d], blabel = g.label[d], bhead = g.head[d], btail = g.tail[d];
    u32 bcnt[SMAX];
#pragma unroll
    for (int t = 0; t < SMAX; ++t) {
      u32 const v = g.cnt[d * S + (t < S ? t : 0)];
      bcnt[t] = t < S ? v : 0u;
    }
    u32 const brole0 = g.role[d * 2], brole1 = g.role[d * 2 + 1];
    u32 const bdesc_self = g.sdesc[d];  // a never-merged node's only slice is itself
    // ---- is_potential_buddy(nid, cand) ----
    if (xn == 1 && bn == 1 && (xe0 >> 2) == d && (be0 >> 2) == nid) break;
    if (bn > 2 || bn == 0) break;
    if ((be0 >> 2) == d || (bn == 2 && (be1 >> 2) == d)) break;
    u32 const expected = mirror_of(nid, cand);
    u32 const exp_src_minus = (expected >> 1) & 1u;
    bool const dir_dflt = (exp_src_minus == 0u) == (bsign != 0);
    u32 const b_exp_minus = dir_dflt ? (bsign ? 0u : 1u) : (bsign ? 1u : 0u);
    bool const m0 = ((be0 >> 1) & 1u) == b_exp_minus;
    bool const m1 = bn == 2 && ((be1 >> 1) & 1u) == b_exp_minus;
    int const c1 = (m0 ? 1 : 0) + (m1 ? 1 : 0);
    int const c2 = static_cast<int>(bn) - c1;
    if (c1 != 1 || c2 != 1) break;
    u32 const f = m0 ? be0 : be1;
    u32 const f2 = m0 ? be1 : be0;
    if (f != expected) break;
    if ((f2 >> 2) == nid) break;
    u32 const fn = f2 >> 2;
    // ---- batch 2: the far neighbour and b's first slice ----
    u32 fnn = g.nedge[fn];
    uint2 const fev = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(fn) * g.ecap);
    u32 const bdesc_head = g.sdesc[bhead != kNoNode ? bhead : d];
    // one wait for the whole batch: without this the loads that are only used after the next test are sunk behind it
    asm volatile("" ::"v"(fnn), "v"(fev.x), "v"(fev.y), "v"(bdesc_head));
    u32 const bdesc = (bhead == d || bhead == kNoNode) ? bdesc_self : bdesc_head;
    if (fnn > 2) break;
    u32 fe0 = fev.x, fe1 = fev.y, fe2 = 0;
    // ---- opposite side of nid ----
    if (copp > 1) break;
    if (copp == 1) {
      if (!opp_known || opp_val != opp) {
        flush();
        opp_ok = is_potential_buddy_f2(g, nid, opp, &opp_f2);
        opp_known = true;
        opp_val = opp;
        opp_nb = opp >> 2;
      }
      if (!opp_ok) break;
      // a cycle folding back onto the opposite side: let the generic code sort it out
      if (d == opp_nb || fn == opp_nb || d == opp_f2 || fn == opp_f2) break;
    }
    if (fn == d || fnn == 0) break;  // cannot happen on a consistent graph
    // ---- Node::Merge(b, kind) + Kmer::Merge ----
    u32 const ob = d, kind = cand & 3u;
    bool const append = kind == 0 || kind == 1;
    bool const rc = kind == 1 || kind == 2;
    if (bhead != kNoNode && bhead == btail) {  // single slice: everything in registers
      u32 st = sd_st(bdesc), ln = sd_ln(bdesc), rcb = sd_rc(bdesc);
      u32 const s = bhead;
      if (rc) {
        st = base_len(g, s) - st - ln;
        rcb ^= 1u;
      }
      if (ln > K1) {
        if (append) st += K1;
        ln -= K1;
        g.sdesc[s] = sd_make(st, ln, rcb);
        if (append) {
          g.snext[xtail] = s;
          g.sprev[s] = xtail;
          g.snext[s] = kNoNode;
          xtail = s;
        } else {
          g.sprev[xhead] = s;
          g.snext[s] = xhead;
          g.sprev[s] = kNoNode;
          xhead = s;
        }
      }
    } else {
      g.head[nid] = xhead;
      g.tail[nid] = xtail;
      if (rc) slices_reverse(g, ob);
      if (append) {
        slices_drop_front(g, ob, K1);
        if (g.head[ob] != kNoNode) {
          g.snext[g.tail[nid]] = g.head[ob];
          g.sprev[g.head[ob]] = g.tail[nid];
          g.tail[nid] = g.tail[ob];
        }
      } else {
        slices_drop_back(g, ob, K1);
        if (g.head[ob] != kNoNode) {
          g.sprev[g.head[nid]] = g.tail[ob];
          g.snext[g.tail[ob]] = g.head[nid];
          g.head[nid] = g.head[ob];
        }
      }
      xhead = g.head[nid];
      xtail = g.tail[nid];
      // settle these loads inside the rare branch: pending at the join they would make the common path wait for
      // all its outstanding memory operations on every merge
      asm volatile("" ::"v"(xhead), "v"(xtail));
    }
    g.head[ob] = kNoNode;
    g.tail[ob] = kNoNode;
    xlen += blen - K1;
    xlabel |= blabel;
    {
      u64 const this_len = xlen, other_len = blen;  // node.cpp:91: length AFTER the merge
#pragma unroll
      for (int t = 0; t < SMAX; ++t)
        if (t < S) xcnt[t] = weighted_floor_avg(xcnt[t], this_len, bcnt[t], other_len);
      xrole0 = weighted_floor_avg(xrole0, this_len, brole0, other_len);
      xrole1 = weighted_floor_avg(xrole1, this_len, brole1, other_len);
    }
    // ---- edges (graph.cpp:600-645) ----
    if (xe0 == cand) xe0 = xe1;  // erase_edge(nid, cand)
    xe1 = 0;
    xn--;
    u32 const src_minus = (kind >> 1) & 1u, dst_minus = kind & 1u;
    u32 const o2n = f2;
    u32 const o2n_src_minus = (o2n >> 1) & 1u, o2n_dst_minus = o2n & 1u;
    u32 const ne_src_minus = (dst_minus != o2n_src_minus) ? (src_minus ^ 1u) : src_minus;
    u32 const s2n = (fn << 2) | (ne_src_minus << 1) | o2n_dst_minus;
    // emplace_edge(nid, s2n): xn is 0 or 1 here
    if (!(xn == 1 && xe0 == s2n)) {
      if (xn == 0) xe0 = s2n; else xe1 = s2n;
      xn++;
    }
    {  // emplace_edge(fn, mirror(nid, s2n)); erase_edge(fn, mirror(ob, o2n))
      u32 const add = mirror_of(nid, s2n), del = mirror_of(ob, o2n);
      bool const present = (fnn >= 1 && fe0 == add) || (fnn >= 2 && fe1 == add);
      if (!present) {
        if (fnn == 0) fe0 = add; else if (fnn == 1) fe1 = add; else fe2 = add;
        fnn++;
      }
      if (fnn >= 1 && fe0 == del) {
        fe0 = fe1;
        fe1 = fe2;
        fe2 = 0;
        fnn--;
      } else if (fnn >= 2 && fe1 == del) {
        fe1 = fe2;
        fe2 = 0;
        fnn--;
      } else if (fnn >= 3 && fe2 == del) {
        fe2 = 0;
        fnn--;
      }
      g.nedge[fn] = static_cast<u8>(fnn);
      g.edge[static_cast<size_t>(fn) * g.ecap] = fe0;
      g.edge[static_cast<size_t>(fn) * g.ecap + 1] = fe1;
      if (fnn > 2) g.edge[static_cast<size_t>(fn) * g.ecap + 2] = fe2;
    }
    CCOUNT(11);
#ifdef MA_PROFILE
    ++dbg_nm;
#endif
    absorbed[ob] = 2;  // absorbed by the register walk: nothing points at it any more
    dirty = true;
  }
#ifdef MA_PROFILE
  { int const ph = g.dbg_phase ? 1 : 0; g.dbg_merges[ph] += dbg_nm; g.dbg_maxwalk[ph] = max(g.dbg_maxwalk[ph], dbg_nm); g.dbg_walks[ph] += dbg_nm ? 1 : 0; }
#endif
  flush();
  compress_node(g, nid, dflt, absorbed);  // re-evaluates from memory: finishes the walk or stops it
}

// Chain-following table in LDS, so that a walk does not touch HBM to find its way.  A walk arrives at node b by an
// edge whose dst-side strand bit is j and leaves by the edge of b whose src-side bit is j (the back edge has !j:
// Kmer sign cancels out of FindEdgesInDirection here).  State = b << 1 | j; next[state] = state after one hop, or
// kNoLink when b does not have exactly two edges.  The table is a HINT: every merge decision is re-derived from the
// records in memory, so a stale entry only ends a lane-parallel run early.
constexpr u32 kLinkCap = 2048;  // nodes covered (2 x u16 each = the kernel's 8 KB of LDS)
constexpr u32 kNoLink = 0xFFFFu;
__device__ __forceinline__ void set_links(Win& g, u32 i, u32 n_edges, u32 e0, u32 e1) {
  if (i >= g.link_cap) return;
  u16* next = reinterpret_cast<u16*>(g.link);
#pragma unroll
  for (u32 j = 0; j < 2; ++j) {
    u32 const on = (((e0 >> 1) & 1u) != j) ? e1 : e0;
    next[i * 2 + j] = static_cast<u16>((n_edges == 2 && (on >> 2) < 0x7FFFu) ? (((on >> 2) << 1) | (on & 1u)) : kNoLink);
  }
}
__device__ __forceinline__ void build_links(Win& g) {
  u32 const top = g.n < g.link_cap ? g.n : g.link_cap;
  for (u32 i = lane_id(); i < top; i += 64) {
    uint2 const ev = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(i) * g.ecap);
    set_links(g, i, g.alive[i] ? g.nedge[i] : 0u, ev.x, ev.y);
  }
  wave_sync_mem(g.lds);
}

// floor(num / den) for num < 2^32 with m = floor(2^32 / den): the estimate mulhi(num, m) is q - 1 or q
__device__ __forceinline__ u32 magic_floor_div(u32 num, u32 den, u32 m) {
  u32 q = __umulhi(num, m);
  u32 const r = num - q * den;
  return q + (r >= den ? 1u : 0u);
}

// The same walk with the 64 lanes put to work.  A chain walk is a strictly sequential recurrence (every merge
// re-links the absorber to the node after the absorbed one), but only two things in it are truly serial: following
// the chain (one 16-byte record per hop) and the floor-rounded running averages of Node::Merge (node.cpp:93-104).
// Everything else -- the buddy predicates, the slice trimming and linking, the bookkeeping of the absorbed node --
// is independent per chain position once the position's incoming edge is known.  So: follow up to 64 hops reading
// records only, hand position t to lane t, evaluate on ORIGINAL records exactly the predicates the sequential walk
// would evaluate on its rewritten ones (the rewrite replaces the back edge mirror(b[t-1], ..) of b[t] by
// mirror(nid, cand[t]): same direction bit, so the tests translate one to one), cut the run at the first position
// with any failing or unusual test, and apply the surviving prefix: slices and flags lane-parallel, the averages as a
// scalar recurrence with a per-position reciprocal prepared in parallel.  Takes only merges the sequential walk
// takes, in the same order; whatever it leaves is re-evaluated from memory by compress_node_fast / compress_node.
__device__ __forceinline__ bool compress_walk_par(Win& g, u32 nid, bool dflt, u8* absorbed) {
  if (static_cast<i64>(nid) == g.source || static_cast<i64>(nid) == g.sink) return true;
  u32 const xn = g.nedge[nid];
  if (xn > 2 || xn == 0) return true;
  u32 const lane = lane_id();
  u32 const K = static_cast<u32>(g.k), K1 = K - 1;
  int const S = g.S;
  u32 xe0, xe1;
  {
    uint2 const ev = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(nid) * g.ecap);
    xe0 = ev.x;
    xe1 = xn == 2 ? ev.y : 0u;
  }
  u32 const xsign = g.sign[nid];
  u32 const exp_minus = dflt ? (xsign ? 0u : 1u) : (xsign ? 1u : 0u);
  if ((xe0 >> 2) == nid || (xn == 2 && (xe1 >> 2) == nid)) return true;
  bool const d0 = ((xe0 >> 1) & 1u) == exp_minus;
  bool const d1 = xn == 2 && ((xe1 >> 1) & 1u) == exp_minus;
  if ((d0 ? 1 : 0) + (d1 ? 1 : 0) != 1) return true;
  u32 cand = d0 ? xe0 : xe1;
  u32 const opp = d0 ? xe1 : xe0;  // the edge on the other side (xn == 2 only): untouched by this walk
  u32 opp_nb = kNoNode, opp_f2 = kNoNode;
  if (xn == 2) {
    if (!is_potential_buddy_f2(g, nid, opp, &opp_f2)) return true;
    opp_nb = opp >> 2;
  }
  bool const append = ((cand >> 1) & 1u) == 0u;  // PLUS_* appends, MINUS_* prepends: constant along the walk
  bool loaded = false;
  bool settled = false;  // the walk ended on a test the generic code fails too: nothing left for it to re-evaluate
  bool clean = true;     // this chunk's rewiring replaced exactly one edge of the node after the run
  u32 xlen = 0, xlabel = 0, xhead = 0, xtail = 0, X0 = 0, X1 = 0, X2 = 0, X3 = 0;
  u32 exp_c = mirror_of(nid, cand);  // the back edge of the first chain node as memory holds it
  while (true) {
    TSTAMP(ts0);
    // ---- follow the chain through the link table: position t -> lane t ----
    u32 my_cand;
    u32 nst = 0;
    {
      const u16* next = reinterpret_cast<const u16*>(g.link);
      // everything but the capture runs on the scalar unit: one LDS read per hop
      u32 st = __builtin_amdgcn_readfirstlane(((cand >> 2) << 1) | (cand & 1u)), my_st = 0;
#pragma nounroll
      while (true) {
        my_st = lane == nst ? st : my_st;
        ++nst;
        if (nst == 64 || (st >> 1) >= g.link_cap) break;
        u32 const nx = __builtin_amdgcn_readfirstlane(static_cast<u32>(next[st]));
        if (nx == kNoLink) break;
        st = nx;
      }
      my_cand = ((my_st >> 1) << 2) | (cand & 2u) | (my_st & 1u);
    }
    TSTAMP(ts1);
    TACC(1, ts0, ts1);
    // ---- every position's tests, on its own lane ----
    bool const act = lane < nst;
    u32 const d = act ? (my_cand >> 2) : nid;
    u32 const bn = g.nedge[d];
    uint2 const bev = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(d) * g.ecap);
    u32 const bsign = g.sign[d];
    u32 const blen = g.len[d], blabel = g.label[d], bhead = g.head[d], btail = g.tail[d];
    u32 const a0 = g.cnt[d * S], a1 = S > 1 ? g.cnt[d * S + 1] : 0u;
    u32 const a2 = g.role[d * 2], a3 = g.role[d * 2 + 1];
    bool ok = act && static_cast<i64>(d) != g.source && static_cast<i64>(d) != g.sink;
    ok = ok && bn == 2 && (bev.x >> 2) != d && (bev.y >> 2) != d;
    u32 const kind = my_cand & 3u;
    u32 f2;
    {
      u32 const exp_src_minus = (kind & 1u) ^ 1u;
      bool const dir_dflt = (exp_src_minus == 0u) == (bsign != 0);
      u32 const b_exp_minus = dir_dflt ? (bsign ? 0u : 1u) : (bsign ? 1u : 0u);
      bool const m0 = ((bev.x >> 1) & 1u) == b_exp_minus, m1 = ((bev.y >> 1) & 1u) == b_exp_minus;
      ok = ok && (m0 != m1);
      u32 const f = m0 ? bev.x : bev.y;
      f2 = m0 ? bev.y : bev.x;
      // the back edge as memory holds it: the mirror of the previous position's onward edge (of nid's for t = 0)
      u32 const back = __shfl_up(mirror_of(d, f2), 1);
      ok = ok && f == (lane == 0 ? exp_c : back) && (f2 >> 2) != nid;
    }
    u32 const fn = ok ? (f2 >> 2) : nid;
    u32 const fnn = g.nedge[fn];
    u32 const sl = (ok && bhead != kNoNode) ? bhead : d;
    u32 const bdesc = g.sdesc[sl];
    bool const neg = act && (!ok || fnn > 2);  // so far only tests of the generic predicates themselves
    ok = ok && fnn <= 2 && fnn != 0 && fn != d;
    ok = ok && !(xn == 2 && (d == opp_nb || fn == opp_nb || d == opp_f2 || fn == opp_f2));
    ok = ok && bhead != kNoNode && bhead == btail;
    u32 const src_minus = (kind >> 1) & 1u, dst_minus = kind & 1u;
    ok = ok && dst_minus == ((f2 >> 1) & 1u);  // the rewired edge leaves nid on the same side
    u32 const s2n = (f2 & ~3u) | (src_minus << 1) | (f2 & 1u);
    ok = ok && !(xn == 2 && s2n == opp);
    ok = ok && blen >= K1;
    bool negv;
    {  // the table's idea of this position's incoming edge must be what the previous position really leads to
      u32 const prev_s2n = __shfl_up(s2n, 1);
      bool const chained = lane == 0 || my_cand == prev_s2n;
      ok = ok && chained;
      negv = neg && chained;
    }
    u32 r;
    {
      unsigned long long const bad = __ballot(!ok);
      r = bad ? static_cast<u32>(__builtin_ctzll(bad)) : 64u;
    }
    // Position r failed.  If it failed a generic predicate, evaluated on what memory holds for it now (position 0:
    // as read; later positions: as read, when the rewiring below swapped exactly one edge, which the back-edge
    // translation already accounts for), the generic code would stop here as well.
    TSTAMP(ts2);
    TACC(2, ts1, ts2);
    bool const neg_r = r < 64 && __builtin_amdgcn_readlane(static_cast<u32>(negv), r & 63u) != 0;
    if (r == 0) {
      settled = neg_r;
      break;
    }
    if (!loaded) {
      // uniform values, told so: the running averages below then run on the scalar unit
      xlen = __builtin_amdgcn_readfirstlane(g.len[nid]);
      xlabel = g.label[nid];
      xhead = g.head[nid];
      xtail = g.tail[nid];
      X0 = __builtin_amdgcn_readfirstlane(g.cnt[nid * S]);
      X1 = S > 1 ? __builtin_amdgcn_readfirstlane(g.cnt[nid * S + 1]) : 0u;
      X2 = __builtin_amdgcn_readfirstlane(g.role[nid * 2]);
      X3 = __builtin_amdgcn_readfirstlane(g.role[nid * 2 + 1]);
    }
    bool in = lane < r;
    // lengths after each merge (inclusive prefix sum), reciprocals, overflow bound of the 32-bit averages
    u32 this_len = in ? blen - K1 : 0u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      u32 const up = __shfl_up(this_len, o);
      if (lane >= static_cast<u32>(o)) this_len += up;
    }
    this_len += xlen;
    u32 const den = this_len + blen;
    {
      u32 mx = in ? max(max(a0, a1), max(a2, a3)) : 0u, md = in ? den : 0u;
      u32 lb = in ? blabel : 0u;
      bool const wide = in && (this_len < xlen || den < this_len);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, static_cast<u32>(__shfl_xor(mx, o)));
        md = max(md, static_cast<u32>(__shfl_xor(md, o)));
        lb |= static_cast<u32>(__shfl_xor(lb, o));
      }
      mx = max(max(mx, max(X0, X1)), max(X2, X3));
      if (__ballot(wide) || ((static_cast<u64>(mx) * md) >> 32) != 0) break;  // 64-bit averages: sequential code
      xlabel |= lb;
    }
    if (!loaded) loaded = true;
    u32 const magic = static_cast<u32>((1ull << 32) / (den < 2 ? 2u : den));
    // ---- the serial part: four floor-rounded running averages ----
    {
      // wave-uniform by construction; said explicitly so that the loop is selected onto the scalar unit (the vector
      // 32-bit multiplies run at quarter rate and this loop is the serial part of the walk)
      u32 s0 = __builtin_amdgcn_readfirstlane(X0), s1 = __builtin_amdgcn_readfirstlane(X1);
      u32 s2 = __builtin_amdgcn_readfirstlane(X2), s3 = __builtin_amdgcn_readfirstlane(X3);
      u32 const rr = __builtin_amdgcn_readfirstlane(r);
#pragma nounroll
      for (u32 t = 0; t < rr; ++t) {
        u32 const tl = __builtin_amdgcn_readlane(this_len, t), ol = __builtin_amdgcn_readlane(blen, t);
        u32 const mg = __builtin_amdgcn_readlane(magic, t), dn = tl + ol;
        s0 = magic_floor_div(s0 * tl + __builtin_amdgcn_readlane(a0, t) * ol, dn, mg);
        s1 = magic_floor_div(s1 * tl + __builtin_amdgcn_readlane(a1, t) * ol, dn, mg);
        s2 = magic_floor_div(s2 * tl + __builtin_amdgcn_readlane(a2, t) * ol, dn, mg);
        s3 = magic_floor_div(s3 * tl + __builtin_amdgcn_readlane(a3, t) * ol, dn, mg);
      }
      X0 = s0;
      X1 = s1;
      X2 = s2;
      X3 = s3;
    }
    xlen = __builtin_amdgcn_readlane(this_len, r - 1);
    // ---- slices: trim k-1 bases off the joining end, then link the survivors in walk order ----
    {
      u32 st = sd_st(bdesc), ln = sd_ln(bdesc), rcb = sd_rc(bdesc);
      if (kind == 1 || kind == 2) {
        st = base_len(g, sl) - st - ln;
        rcb ^= 1u;
      }
      bool const keep = in && ln > K1;
      if (keep) {
        if (append) st += K1;
        ln -= K1;
        g.sdesc[bhead] = sd_make(st, ln, rcb);
      }
      unsigned long long const kept = __ballot(keep);
      unsigned long long const below = kept & ((1ull << lane) - 1ull), above = kept & ~((2ull << lane) - 1ull);
      u32 const pl = below ? 63u - static_cast<u32>(__builtin_clzll(below)) : 0u;
      u32 const nl = above ? static_cast<u32>(__builtin_ctzll(above)) : 0u;
      u32 const ps = __shfl(bhead, static_cast<int>(pl)), ns = __shfl(bhead, static_cast<int>(nl));
      if (keep) {
        if (append) {
          u32 const before = below ? ps : xtail;
          g.snext[before] = bhead;
          g.sprev[bhead] = before;
          g.snext[bhead] = above ? ns : kNoNode;
        } else {
          u32 const after = below ? ps : xhead;
          g.sprev[after] = bhead;
          g.snext[bhead] = after;
          g.sprev[bhead] = above ? ns : kNoNode;
        }
      }
      if (kept) {
        u32 const last = __builtin_amdgcn_readlane(bhead, 63u - static_cast<u32>(__builtin_clzll(kept)));
        if (append) xtail = last; else xhead = last;
      }
      if (in) {
        g.head[d] = kNoNode;
        g.tail[d] = kNoNode;
        absorbed[d] = 2;
      }
    }
    // ---- edges: nid now points at the node after the last absorbed one (graph.cpp:600-645) ----
    u32 const s2n_last = __builtin_amdgcn_readlane(s2n, r - 1);
    u32 const f2_last = __builtin_amdgcn_readlane(f2, r - 1);
    u32 const ob_last = __builtin_amdgcn_readlane(d, r - 1);
    if (xn == 2) {
      xe0 = opp;
      xe1 = s2n_last;
    } else {
      xe0 = s2n_last;
      xe1 = 0;
    }
    {
      u32 const fl = f2_last >> 2;
      u32 fnl = g.nedge[fl];
      uint2 const fev = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(fl) * g.ecap);
      u32 fe0 = fev.x, fe1 = fev.y, fe2 = 0;
      u32 const add = mirror_of(nid, s2n_last), del = mirror_of(ob_last, f2_last);
      bool const present = (fnl >= 1 && fe0 == add) || (fnl >= 2 && fe1 == add);
      u32 const fnl_before = fnl;
      if (!present) {
        if (fnl == 0) fe0 = add; else if (fnl == 1) fe1 = add; else fe2 = add;
        fnl++;
      }
      if (fnl >= 1 && fe0 == del) {
        fe0 = fe1;
        fe1 = fe2;
        fe2 = 0;
        fnl--;
      } else if (fnl >= 2 && fe1 == del) {
        fe1 = fe2;
        fe2 = 0;
        fnl--;
      } else if (fnl >= 3 && fe2 == del) {
        fe2 = 0;
        fnl--;
      }
      clean = !present && fnl == fnl_before;
      g.nedge[fl] = static_cast<u8>(fnl);
      g.edge[static_cast<size_t>(fl) * g.ecap] = fe0;
      g.edge[static_cast<size_t>(fl) * g.ecap + 1] = fe1;
      if (fnl > 2) g.edge[static_cast<size_t>(fl) * g.ecap + 2] = fe2;
      set_links(g, fl, fnl, fe0, fe1);
    }
    cand = s2n_last;
    exp_c = mirror_of(nid, cand);
    TSTAMP(ts3);
    TACC(3, ts2, ts3);
    if (r < 64) {
      settled = neg_r && clean;
      break;  // the exit below fences
    }
    wave_sync_mem(g.lds);  // the next chunk links its first surviving slice behind this chunk's last one
  }
  if (loaded) {
    g.edge[static_cast<size_t>(nid) * g.ecap] = xe0;
    g.edge[static_cast<size_t>(nid) * g.ecap + 1] = xe1;
    g.len[nid] = xlen;
    g.label[nid] = static_cast<u8>(xlabel);
    g.cnt[nid * S] = X0;
    if (S > 1) g.cnt[nid * S + 1] = X1;
    g.role[nid * 2] = X2;
    g.role[nid * 2 + 1] = X3;
    g.head[nid] = xhead;
    g.tail[nid] = xtail;
    set_links(g, nid, 0u, 0u, 0u);  // several slices now: no lane-parallel walk passes through it
    wave_sync_mem(g.lds);
  }
  return settled;
}

// Exact negative filter for CompressNode, cheap enough to run on 64 nodes at once: find_compressible_edge(i, .) can
// only succeed if every neighbour of i passes is_potential_buddy, and that needs the neighbour to have exactly two
// edges, one of them back to i, the other to a node with at most two edges.  False means "no direction of i is
// compressible on what memory holds now"; true means "ask the real predicates".
__device__ __forceinline__ bool may_compress(const Win& g, u32 i) {
  if (static_cast<i64>(i) == g.source || static_cast<i64>(i) == g.sink) return false;
  u32 const xn = g.nedge[i];
  if (xn == 0 || xn > 2) return false;
  uint2 const ev = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(i) * g.ecap);
  for (u32 x = 0; x < xn; ++x) {
    u32 const nb = (x == 0 ? ev.x : ev.y) >> 2;
    if (g.nedge[nb] != 2) return false;
    uint2 const ne = *reinterpret_cast<const uint2*>(g.edge + static_cast<size_t>(nb) * g.ecap);
    bool const via0 = (ne.x >> 2) == i && g.nedge[ne.y >> 2] <= 2;
    bool const via1 = (ne.y >> 2) == i && g.nedge[ne.x >> 2] <= 2;
    if (!via0 && !via1) return false;
  }
  return true;
}

// CompressGraph (graph.cpp:558-576)
__device__ __forceinline__ void compress_graph(Win& g, u32 comp) {
  TSTAMP(tg0);
  u8* absorbed = reinterpret_cast<u8*>(g.scratch + g.nc);
  for (u32 i = lane_id(); i < g.n; i += 64) absorbed[i] = 0;
  wave_sync_mem(g.lds);
  if (g.S <= 2) build_links(g);
  TSTAMP(tg2);
  TACC(5, tg0, tg2);
  for (u32 base = 0; base < g.n; base += 64) {
    // alive / comp do not change inside this loop; absorbed does and is re-read when a node's turn comes
    u32 const il = base + lane_id();
    unsigned long long todo = __ballot(il < g.n && g.alive[il] && g.comp[il] == comp && !absorbed[il]);
    while (todo) {
      // Nodes that cannot compress leave the graph as it is, so one lane-parallel look at the rest of the block
      // stays valid up to the first node that may; after that node has had its turn the rest is looked at again.
      TSTAMP(tf0);
      bool const pre = ((todo >> lane_id()) & 1ull) && !absorbed[il] && may_compress(g, il);
      unsigned long long const can = __ballot(pre) & todo;
      TSTAMP(tf1);
      TACC(0, tf0, tf1);
      if (!can) break;
      u32 const first = static_cast<u32>(__builtin_ctzll(can));
      u32 const i = base + first;
      todo &= ~((2ull << first) - 1ull);
#pragma nounroll
      for (int dir = 1; dir >= 0; --dir) {
        if (g.S <= 2) {
          if (!compress_walk_par(g, i, dir != 0, absorbed)) compress_node_fast<2>(g, i, dir != 0, absorbed);
        }
        else compress_node_fast<kMaxSamples>(g, i, dir != 0, absorbed);
      }
    }
  }
  // absorbed == 2 (register walk): remove_node would only look for mirror edges that the walk has already
  // erased (no edge to an absorbed node is ever created again: rewired edges point at absorbers), so the
  // node is simply switched off.  absorbed == 1 (generic path) goes through remove_node, in index order.
  bool any_generic = false;
  for (u32 i = lane_id(); i < g.n; i += 64) {
    u32 const ab = absorbed[i];
    if (ab == 2) {
      g.alive[i] = 0;
      g.nedge[i] = 0;
    }
    any_generic |= ab == 1;
  }
  wave_sync_mem(g.lds);
  if (__ballot(any_generic))
    for (u32 i = 0; i < g.n; ++i)
      if (absorbed[i] == 1) remove_node(g, i);
  TSTAMP(tg1);
  TACC(4, tg0, tg1);
}

// RemoveTips (graph.cpp:801-840): one round of tip collection, in index order; the caller removes them and
// re-compresses until a round finds none
__device__ __forceinline__ u32 collect_tips(Win& g, u32 comp, u32* rm) {
  u32 nrm = 0;
  for (u32 base = 0; base < g.n; base += 64) {
    u32 const i = base + lane_id();
    bool flag = false;
    if (i < g.n && g.alive[i]) {
      bool const anchor = static_cast<i64>(i) == g.source || static_cast<i64>(i) == g.sink;
      if (g.comp[i] == comp && !anchor && g.nedge[i] <= 1) {
        u32 const uniq = g.len[i] - static_cast<u32>(g.k) + 1;
        flag = uniq < static_cast<u32>(g.k);
      }
    }
    ordered_append(flag, i, rm, nrm);
  }
  wave_sync_mem(g.lds);
  return nrm;
}

// ---- sequence spelling ----
__device__ __forceinline__ u8 canon_base(const Win& g, u32 o, u32 x) {  // x-th base of slice o's base string
  u32 const sv = g.bsrc[o];
  const u8* p = (sv & 0x80000000u) ? g.readb + (sv & 0x3FFFFFFFu) : ((sv & 0x40000000u) ? g.pool + (sv & 0x3FFFFFFFu) : g.refb + sv);
  return g.bsign[o] ? p[x] : dev_complement(p[base_len(g, o) - 1 - x]);
}
__device__ __forceinline__ u8 slice_base(const Win& g, u32 o, u32 d, u32 j) {  // j-th base of slice (o, d)
  u32 const st = sd_st(d), rc = sd_rc(d);
  u32 const pp = st + j;
  return rc ? dev_complement(canon_base(g, o, base_len(g, o) - 1 - pp)) : canon_base(g, o, pp);
}
// append the oriented sequence of `node` (dflt: stored orientation, else reverse complement) minus its
// first `skip` bases to out[*pos..], bounded by cap.  Returns false on overflow.
// A unitig is a linked list of slices, mostly one base each (every merged k-mer adds its non-overlapping base), so
// spelling it base by base is one chain of dependent loads per base.  Instead the list is walked once for the slice
// ids (one dependent load per slice), and the slices are then spelled 64 at a time: descriptor, source and base
// loads of 64 slices are in flight together and a prefix sum of the slice lengths gives every base its position.
// Slice lists are linked lists through HBM, and a merged chain of k-mers is a list of single-base slices in k-mer
// index order, i.e. scattered: following one costs a memory round trip per base.  Instead every slice learns its
// list's last slice and its distance from it by pointer jumping (Wyllie), all lists of the window at once, in LDS:
// rank[s] = dist << 16 | last.  A slice that was trimmed away (slices_drop_front) still points INTO a live list;
// such a stale link is recognised by the missing back pointer and starts a list of its own.
__device__ __forceinline__ void rank_slices(Win& g) {
  u32 const lane = lane_id();
  u32* rank = g.link;
  for (u32 i = lane; i < g.n; i += 64) {
    u32 const nx = g.snext[i];
    bool const fwd = nx != kNoNode && nx < g.n && g.sprev[nx] == i;
    rank[i] = fwd ? ((1u << 16) | nx) : i;
  }
  wave_sync_mem(g.lds);
  while (true) {
    bool changed = false;
    for (u32 i = lane; i < g.n; i += 64) {
      u32 const mine = rank[i];
      u32 const to = rank[mine & 0xFFFFu];
      if ((to & 0xFFFFu) != (mine & 0xFFFFu)) {  // my target is not a last slice yet: jump over it
        rank[i] = ((mine & 0xFFFF0000u) + (to & 0xFFFF0000u)) | (to & 0xFFFFu);
        changed = true;
      }
    }
    wave_sync_mem(g.lds);
    if (!__ballot(changed)) break;
  }
}

__device__ __forceinline__ bool emit_node_seq(const Win& g, u32 node, bool dflt, u32 skip, u8* out, u32* pos, u32 cap) {
  u32* ids = g.scratch + 7u * g.nc;  // adjacency scratch of the traversal index: dead once the walks are enumerated
  u32 const lane = lane_id();
  // The list is followed 64 candidates at a time: slices of neighbouring k-mers have neighbouring indices, so lane l
  // looks at slice s + l (s - l when walking backwards); as long as each one links to the next index the whole run
  // belongs to the list -- by induction from s -- and costs one round trip instead of one per slice.
  u32 ns = 0;
  if (g.ranked) {
    // every slice of the window looks itself up: on this node's list (same last slice, not beyond its head)?
    u32 const h = g.head[node], t = g.tail[node];
    if (h != kNoNode) {
      u32 const hd = g.link[h] >> 16;
      for (u32 i = lane; i < g.n; i += 64) {
        u32 const rk = g.link[i];
        if ((rk & 0xFFFFu) == t && (rk >> 16) <= hd) ids[dflt ? hd - (rk >> 16) : (rk >> 16)] = i;
      }
      ns = hd + 1u;
    }
  } else {
    u32 s = dflt ? g.head[node] : g.tail[node];
    while (s != kNoNode) {
      u32 const mine = dflt ? s + lane : s - lane;  // wraps past 0 when walking backwards: rejected by the bound
      u32 const nx = mine < g.n ? (dflt ? g.snext[mine] : g.sprev[mine]) : kNoNode;
      bool const linked = nx == (dflt ? mine + 1u : mine - 1u) && (dflt || mine != 0u);
      unsigned long long const m = __ballot(linked);
      u32 const run = m == ~0ull ? 63u : static_cast<u32>(__builtin_ctzll(~m));  // lanes 0 .. run are on the list
      if (lane <= run) ids[ns + lane] = mine;
      ns += run + 1u;
      s = static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(nx), static_cast<int>(run)));
    }
  }
  wave_sync_mem(g.lds);
  bool over = false;
  u32 emitted = 0;  // bases of the node spelled by the blocks before this one
  for (u32 b0 = 0; b0 < ns; b0 += 64) {
    u32 const idx = b0 + lane;
    u32 s = 0, d = 0, ln = 0, bl = 0, bs = 1;
    const u8* bp = g.refb;
    if (idx < ns) {
      s = ids[idx];
      d = g.sdesc[s];
      ln = sd_ln(d);
      u32 const sv = g.bsrc[s];
      bp = (sv & 0x80000000u) ? g.readb + (sv & 0x3FFFFFFFu) : ((sv & 0x40000000u) ? g.pool + (sv & 0x3FFFFFFFu) : g.refb + sv);
      bl = base_len(g, s);
      bs = g.bsign[s];
    }
    // j-th base of the slice in ITS orientation: base string position st + j, read backwards and complemented once for a
    // reverse-complemented slice and once more for a base string stored on the other strand
    auto base_of = [](const u8* p, u32 blen, u32 bsign, u32 dd, u32 j) -> u8 {
      u32 pp = sd_st(dd) + j;
      bool comp = false;
      if (sd_rc(dd)) {
        pp = blen - 1 - pp;
        comp = !comp;
      }
      if (!bsign) {
        pp = blen - 1 - pp;
        comp = !comp;
      }
      u8 const c = p[pp];
      return comp ? dev_complement(c) : c;
    };
    u32 inc = ln, mx = ln;
#pragma unroll
    for (u32 o = 1; o < 64; o <<= 1) {
      u32 const y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
      mx = max(mx, static_cast<u32>(__shfl_xor(mx, o, 64)));
    }
    u32 const off = emitted + inc - ln;
    if (mx <= 4u) {  // k-mer sized slices (the raw graph: mostly one base each): a lane per slice
      for (u32 j = 0; j < ln; ++j) {
        u32 const e = off + j;  // index of this base within the node's oriented sequence
        if (e < skip) continue;
        u32 const at = *pos + (e - skip);
        u8 const base = dflt ? base_of(bp, bl, bs, d, j) : dev_complement(base_of(bp, bl, bs, d, ln - 1 - j));
        if (at < cap) out[at] = base; else over = true;
      }
    } else {  // long slices (merged strings of the compact graph): one slice at a time, a lane per base
      u32 const cntb = ns - b0 < 64u ? ns - b0 : 64u;
      for (u32 t = 0; t < cntb; ++t) {
        u32 const dt = __builtin_amdgcn_readlane(d, t), lt = __builtin_amdgcn_readlane(ln, t), ot = __builtin_amdgcn_readlane(off, t);
        u32 const blt = __builtin_amdgcn_readlane(bl, t), bst = __builtin_amdgcn_readlane(bs, t);
        u64 const pa = reinterpret_cast<u64>(bp);
        u32 const plo = __builtin_amdgcn_readlane(static_cast<u32>(pa), t), phi = __builtin_amdgcn_readlane(static_cast<u32>(pa >> 32), t);
        const u8* pt = reinterpret_cast<const u8*>((static_cast<u64>(phi) << 32) | plo);
        for (u32 j = lane; j < lt; j += 64) {
          u32 const e = ot + j;
          if (e < skip) continue;
          u32 const at = *pos + (e - skip);
          u8 const base = dflt ? base_of(pt, blt, bst, dt, j) : dev_complement(base_of(pt, blt, bst, dt, lt - 1 - j));
          if (at < cap) out[at] = base; else over = true;
        }
      }
    }
    emitted += __shfl(inc, 63, 64);
  }
  *pos += emitted > skip ? emitted - skip : 0u;
  wave_sync_mem(g.lds);
  return __ballot(over) == 0;
}

struct OnlineStats {  // base/compute_stats.h:75-125
  u32 n = 0;
  f64 m1 = 0.0, m2 = 0.0;
  __device__ void add(f64 v) {
    u32 const old_n = n++;
    f64 const delta = v - m1;
    f64 const nd = delta / static_cast<f64>(n);
    m1 += nd;
    m2 += (delta * nd * static_cast<f64>(old_n));
  }
  __device__ f64 variance() const { return n < 2 ? 0.0 : m2 / static_cast<f64>(n - 1); }
  __device__ f64 sd() const { return sqrt(variance()); }
};

// ascending sort of a short list kept in the window's scratch: up to 64 values are ranked in registers (lane l counts
// the values that sort before its own; equal values keep their order) instead of shuffling them through memory
__device__ __forceinline__ void isort_u32(u32* v, u32 n);
__device__ __forceinline__ void sort_small_u32(u32* v, u32 n, bool lds_only) {
  if (n > 64) {
    isort_u32(v, n);
    return;
  }
  u32 const lane = lane_id();
  u32 const x = lane < n ? v[lane] : 0xFFFFFFFFu;
  u32 rank = 0;
  for (u32 j = 0; j < n; ++j) {
    u32 const y = __builtin_amdgcn_readlane(x, j);
    rank += (y < x || (y == x && j < lane)) ? 1u : 0u;
  }
  wave_sync_mem(lds_only);  // every lane has read before any lane writes
  if (lane < n) v[rank] = x;
  wave_sync_mem(lds_only);
}
__device__ __forceinline__ void isort_u32(u32* v, u32 n) {
  for (u32 i = 1; i < n; ++i) {
    u32 const x = v[i];
    u32 j = i;
    while (j > 0 && v[j - 1] > x) {
      v[j] = v[j - 1];
      --j;
    }
    v[j] = x;
  }
}
__device__ __forceinline__ u32 median_sorted(const u32* v, u32 n) {  // compute_stats.h:146-159 on sorted data
  if (n == 0) return 0;
  if (n == 1) return v[0];
  u32 const half = v[n / 2];
  if (n % 2 == 1) return half;
  return (half + v[n / 2 - 1]) / 2;
}

}  // namespace

#ifdef MA_PROFILE
#define CPROF_T0() unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define CPROF_ACC(slot)                                                       \
  do {                                                                        \
    unsigned long long _t1 = __builtin_amdgcn_s_memtime();                    \
    if (threadIdx.x == 0) atomicAdd(&g_cprof[slot], _t1 - _t0);               \
    g.dbg_ph[slot] += _t1 - _t0;                                              \
    _t0 = _t1;                                                                \
  } while (0)
extern "C" void ma_debug_cwin(unsigned long long* out, int n) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cwin), sizeof(unsigned long long) * 4 * n);
}
extern "C" void ma_debug_ctime(unsigned long long* out, int n) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ctime), sizeof(unsigned long long) * 6 * n);
}
extern "C" void ma_debug_cmerge(unsigned* out, int n) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cmerge), sizeof(unsigned) * 8 * n);
}
extern "C" void ma_debug_cprof(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cprof), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cprof), z, sizeof(z));
  }
}
#define CSUB_T0() unsigned long long _u0 = __builtin_amdgcn_s_memtime()
#define CSUB_ACC(slot)                                                        \
  do {                                                                        \
    unsigned long long _u1 = __builtin_amdgcn_s_memtime();                    \
    if (threadIdx.x == 0) atomicAdd(&g_cprof[slot], _u1 - _u0);               \
    _u0 = _u1;                                                                \
  } while (0)
#else
#define CPROF_T0() do {} while (0)
#define CPROF_ACC(slot) do {} while (0)
#define CSUB_T0() do {} while (0)
#define CSUB_ACC(slot) do {} while (0)
#endif

// MarkConnectedComponents (graph.cpp:392-463), lane-parallel; returns the number of components.  `lab` is working
// storage for n words: the caller passes LDS when the graph fits (the pointer-jumping loads are dependent gathers),
// the window's scratch otherwise.
__device__ __forceinline__ u32 label_components(Win& g, u32* lab, u32* cid, u32 lane) {
  u32 ncomp_all = 0;
  for (u32 i = lane; i < g.n; i += 64) lab[i] = i;
  wave_sync_mem(g.lds);
  // FastSV-style hooking (Zhang, Azad, Hu 2020): lab[] is a forest of pointers towards smaller indices;
  // every edge hooks the parent of one end (and the end itself) onto the grandparent of the other, then
  // every node shortcuts to its grandparent.  At the fixed point every tree is a star rooted at the
  // smallest index of its component.
  while (true) {
    bool changed = false;
    for (u32 i = lane; i < g.n; i += 64) {
      u32 const pu = lab[i];
      u32 const gu = lab[pu];
      u32 const ne = g.nedge[i];
      u32 best = gu;
      for (u32 x = 0; x < ne; ++x) {
        u32 const v = g.edge[i * g.ecap + x] >> 2;
        best = min(best, lab[lab[v]]);
      }
      if (best < gu) {
        atomicMin(&lab[pu], best);
        atomicMin(&lab[i], best);
        changed = true;
      }
    }
    wave_sync_mem(g.lds);
    for (u32 i = lane; i < g.n; i += 64) {
      u32 const pu = lab[i];
      u32 const gu = lab[pu];
      if (gu < pu) {
        lab[i] = gu;
        changed = true;
      }
    }
    wave_sync_mem(g.lds);
    if (!__ballot(changed)) break;
  }
  for (u32 base = 0; base < g.n; base += 64) {
    u32 const i = base + lane;
    bool const root = i < g.n && lab[i] == i;
    unsigned long long const m = __ballot(root);
    if (root) cid[i] = ncomp_all + 1 + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)));
    ncomp_all += static_cast<u32>(__popcll(m));
  }
  wave_sync_mem(g.lds);
  for (u32 i = lane; i < g.n; i += 64) g.comp[i] = cid[lab[i]];
  wave_sync_mem(g.lds);
  return ncomp_all;
}

struct CleanArgs {
  DBatch b;
  GraphWs ws;
  ma_asm_out_t out;
  ma_params_t prm;
};

// LDS of the one-wavefront kernels: the shared 8 KB table plus everything that used to be a dynamically indexed local
// (candidate components, accepted walks, the edge order of EnqueueOutgoingEdges): as private arrays those lived in
// scratch memory -- an HBM round trip per access of an already serial loop, 1.6 KB per lane.
constexpr int kMaxCand = 16;
template <u32 kLink>
struct CleanLdsT {
  u32 link[kLink];
  u32 cand_comp[kMaxCand], cand_size[kMaxCand], cand_src[kMaxCand], cand_snk[kMaxCand], cand_soff[kMaxCand], cand_koff[kMaxCand];
  u32 walk_off[kMaxWalks], walk_len[kMaxWalks], walk_minw[kMaxWalks];
  int order[kMaxWalks];
  u32 eq_idx[kEdgeCap], eq_conf[kEdgeCap];
};

using CleanLds = CleanLdsT<kLinkCap>;
template <class SH>
__device__ __forceinline__ void clean_candidates(CleanArgs const& A, Win& g, SH& sh, int a, int w, int ncand, int first_phase, u32 NC);

__global__ __launch_bounds__(64, 2) void k_clean(CleanArgs A) {  // (the rare route: registers rather than occupancy, no spills)
  int const a = blockIdx.x;
  u32 const lane = threadIdx.x;
  GraphWs const& ws = A.ws;
  int const w = static_cast<int>(ws.active[a]);
  ma_params_t const& P = A.prm;
  size_t const nb = static_cast<size_t>(a) * ws.nc;
  if (ws.cg_state && ws.cg_state[a] != 0u) return;  // k_clean_chains + k_clean_tail have this window

  if (ws.win_flags[w] & 4u) {  // build-stage capacity overflow: report and stop retrying
    A.out.win_status[w] = MA_W_TABLE_OVERFLOW | MA_W_NO_HAPLOTYPE;
    A.out.win_ncomp[w] = 0;
    if (lane == 0) atomicOr(&ws.win_flags[w], 1u);
    return;
  }

  __shared__ CleanLds sh;
  u32* const l_link = sh.link;
  Win g;
  g.link = l_link;
  g.ranked = false;
  g.refb = A.b.ref_bases + A.b.ref_off[w];
  g.readb = A.b.read_bases + A.b.read_off[A.b.read_win_off[w]];
  g.ref_len = A.b.ref_off[w + 1] - A.b.ref_off[w];
  g.k = win_kmer(ws, w);
  g.S = ws.num_samples;
  g.n = ws.n_nodes[a];
  g.nc = ws.nc;
  g.min_node_cov = P.min_node_cov;
  g.min_anchor_cov = P.min_anchor_cov;
  g.cnt = ws.nd_cnt + nb * g.S;
  g.role = ws.nd_role + nb * 2;
  g.src = ws.nd_src + nb;
  g.label = ws.nd_label + nb;
  g.sign = ws.nd_sign + nb;
  g.nedge = ws.nd_nedge + nb;
  g.edge = ws.nd_edge + nb * kEdgeCap;
  g.comp = ws.nd_comp + nb;
  g.len = ws.nd_len + nb;
  g.alive = ws.nd_alive + nb;
  g.head = ws.nd_head + nb;
  g.tail = ws.nd_tail + nb;
  g.snext = ws.sl_next + nb;
  g.sprev = ws.sl_prev + nb;
  g.sdesc = ws.sl_desc + nb;
  g.bsrc = g.src;
  g.blen = nullptr;
  g.bsign = g.sign;
  g.pool = nullptr;
  g.ecap = kEdgeCap;
  g.lds = false;
  g.ek = 4;
  g.wk = 7;
  g.pieces = nullptr;
  g.piece_cap = 0;
  g.link_cap = kLinkCap;
  g.arena = ws.arena + static_cast<size_t>(a) * ws.ac;
  g.ac = ws.ac;
  g.scratch = ws.scratch + nb * 32;
  g.source = g.sink = -1;
  g.flags = 0;
  g.dbg_why = 0;
#ifdef MA_PROFILE
  for (int q = 0; q < 6; ++q) g.dbg_t[q] = 0;
  for (int q = 0; q < 16; ++q) g.dbg_ph[q] = 0;
  g.dbg_phase = 0; g.dbg_merges[0] = g.dbg_merges[1] = g.dbg_maxwalk[0] = g.dbg_maxwalk[1] = g.dbg_walks[0] = g.dbg_walks[1] = 0;
#endif
  u32 const NC = ws.nc;
  u32 const K = static_cast<u32>(g.k);
  CPROF_T0();
#ifdef MA_PROFILE
  unsigned long long const t_begin = __builtin_amdgcn_s_memtime();
  u32 dbg_rounds = 0;
#endif

  for (u32 i = lane; i < g.n; i += 64) {
    g.comp[i] = 0;
    g.len[i] = K;
    g.alive[i] = 1;
    g.head[i] = g.tail[i] = i;
    g.snext[i] = g.sprev[i] = kNoNode;
    g.sdesc[i] = sd_make(0u, K, 0u);
  }
  wave_sync_mem(g.lds);

  CPROF_ACC(0);
  // ---- MarkConnectedComponents (graph.cpp:392-463): ids in discovery order over canonical order ----
  // The reference labels components in BFS discovery order starting from the smallest unvisited node, i.e.
  // component ids increase with the components' smallest node index.  Computed lane-parallel: every node
  // carries the smallest node index it knows of in its component (min over neighbours + pointer jumping
  // until nothing changes: O(log n) rounds on the chains of a k-mer graph), then the roots are numbered
  // in index order.
  u32 const ncomp_all = g.n <= kLinkCap ? label_components(g, l_link, g.scratch + NC, lane)
                                        : label_components(g, g.scratch, g.scratch + NC, lane);
  CPROF_ACC(1);
  // component sizes + anchors in one pass (FindSource / FindSink, graph.cpp:469-509): components are
  // disjoint and pruning one never touches another, so the candidates can be resolved up front.
  // Candidate = component with source != sink and ref anchor length >= min_anchor_len.
  const u32* refn = ws.ref_node + static_cast<size_t>(a) * ws.ref_stride;
  u32 const n_refk = g.ref_len >= K + 1 ? g.ref_len - K + 1 : 0;
  u32* const cand_comp = sh.cand_comp;
  u32* const cand_size = sh.cand_size;
  u32* const cand_src = sh.cand_src;
  u32* const cand_snk = sh.cand_snk;
  u32* const cand_soff = sh.cand_soff;
  u32* const cand_koff = sh.cand_koff;
  int ncand = 0;
  {
    // first / last qualifying reference k-mer per component, discovered in reference order
    u32* first_off = g.scratch;            // [ncomp_all+1]
    u32* last_off = g.scratch + NC;        // [ncomp_all+1]
    u32* csize = g.scratch + 2 * NC;       // [ncomp_all+1]
    for (u32 c = lane; c <= ncomp_all; c += 64) {
      first_off[c] = kNoNode;  // minimum of the qualifying offsets
      last_off[c] = 0;         // maximum (only read when first_off exists)
      csize[c] = 0;
    }
    wave_sync_mem(g.lds);
    // neighbouring indices mostly share a component: one atomic per distinct component of the 64, issued by the
    // group's first lane, instead of 64 atomics queueing on one address
    for (u32 base = 0; base < g.n; base += 64) {
      u32 const i = base + lane;
      u32 const c = i < g.n ? g.comp[i] : 0u;
      unsigned long long rem = __ballot(i < g.n);
      while (rem) {
        u32 const l0 = static_cast<u32>(__builtin_ctzll(rem));
        u32 const c0 = __builtin_amdgcn_readlane(c, l0);
        unsigned long long const m = __ballot(i < g.n && c == c0) & rem;
        if (lane == l0) atomicAdd(&csize[c0], static_cast<u32>(__popcll(m)));
        rem &= ~m;
      }
    }
    for (u32 base = 0; base < n_refk; base += 64) {
      u32 const r = base + lane;
      u32 const nd = r < n_refk ? refn[r] : kNoNode;
      bool const valid = nd != kNoNode && nd_total(g, nd) >= g.min_anchor_cov;
      u32 const c = valid ? g.comp[nd] : 0u;
      unsigned long long rem = __ballot(valid);
      while (rem) {
        u32 const l0 = static_cast<u32>(__builtin_ctzll(rem));
        u32 const c0 = __builtin_amdgcn_readlane(c, l0);
        unsigned long long const m = __ballot(valid && c == c0) & rem;
        if (lane == l0) {  // offsets grow with the lane: the group's extremes are its first and last lane
          atomicMin(&first_off[c0], base + l0);
          atomicMax(&last_off[c0], base + 63u - static_cast<u32>(__builtin_clzll(m)));
        }
        rem &= ~m;
      }
    }
    wave_sync_mem(g.lds);
    for (u32 c = 1; c <= ncomp_all; ++c) {
      if (first_off[c] == kNoNode) continue;
      u32 const so = first_off[c], ko = last_off[c];
      if (refn[so] == refn[ko]) continue;                        // same node (graph.cpp:160)
      if (ko - so + K < static_cast<u32>(P.min_anchor_len)) continue;  // graph.cpp:167-173
      if (ncand < kMaxCand) {
        cand_comp[ncand] = c;
        cand_size[ncand] = csize[c];
        cand_src[ncand] = refn[so];
        cand_snk[ncand] = refn[ko];
        cand_soff[ncand] = so;
        cand_koff[ncand] = ko;
        ncand++;
      } else {
        g.flags |= 4u;
      }
    }
    // stable order: size descending, then component id ascending (canonical form of graph.cpp:441)
    for (int i = 1; i < ncand; ++i) {
      int j = i;
      while (j > 0 && cand_size[j - 1] < cand_size[j]) {
        u32 t;
        t = cand_comp[j]; cand_comp[j] = cand_comp[j - 1]; cand_comp[j - 1] = t;
        t = cand_size[j]; cand_size[j] = cand_size[j - 1]; cand_size[j - 1] = t;
        t = cand_src[j]; cand_src[j] = cand_src[j - 1]; cand_src[j - 1] = t;
        t = cand_snk[j]; cand_snk[j] = cand_snk[j - 1]; cand_snk[j - 1] = t;
        t = cand_soff[j]; cand_soff[j] = cand_soff[j - 1]; cand_soff[j - 1] = t;
        t = cand_koff[j]; cand_koff[j] = cand_koff[j - 1]; cand_koff[j - 1] = t;
        --j;
      }
    }
  }

  CPROF_ACC(2);
  clean_candidates(A, g, sh, a, w, ncand, 0, NC);
}

// The candidate components in order (graph.cpp:142-235): PruneComponent, BuildTraversalIndex, HasCycle, complexity gate,
// MaxFlow::NextPath loop, BuildHaplotypes; then the window's status.  first_phase = 0 on the raw graph; 1 when
// k_clean_chains has already done the first CompressGraph (the graph in `g` is then the compact one).
template <class SH>
__device__ __forceinline__ void clean_candidates(CleanArgs const& A, Win& g, SH& sh, int a, int w, int ncand, int first_phase, u32 NC) {
  GraphWs const& ws = A.ws;
  ma_params_t const& P = A.prm;
  u32 const lane = threadIdx.x;
  int const MC = P.max_comps, MH = P.max_haps, ML = P.max_hap_len, MR = P.max_runs;
  u32 const K = static_cast<u32>(g.k);
  u32* const l_link = sh.link;
  u32* const cand_comp = sh.cand_comp;
  u32* const cand_src = sh.cand_src;
  u32* const cand_snk = sh.cand_snk;
  u32* const cand_soff = sh.cand_soff;
  u32* const cand_koff = sh.cand_koff;
  CPROF_T0();
#ifdef MA_PROFILE
  unsigned long long const t_begin = __builtin_amdgcn_s_memtime();
  u32 dbg_rounds = 0;
#endif
  u32 status = 0;
  u32 ncomp_out = 0, slot = 0;
  bool retry = false;
  bool const compact = first_phase != 0;

  for (int ci = 0; ci < ncand && !retry; ++ci) {
    u32 const comp = cand_comp[ci];
    g.ranked = false;  // the table is the chain-following table again
    g.source = cand_src[ci];
    g.sink = cand_snk[ci];
    u32 const anchor_len = cand_koff[ci] - cand_soff[ci] + K;
    const u8* ref_anchor = g.refb + cand_soff[ci];

    // ---- PruneComponent (graph.cpp:515-540) ----
    // compress; remove low coverage; compress; { remove tips; compress } until no tip is left -- written as
    // one loop so that the (large, fully inlined) compression code exists once
#pragma nounroll
    for (int phase = first_phase;; ++phase) {
      if (phase == 1) remove_low_cov(g, comp);
      if (phase >= 2) {
        u32* rm = g.scratch;
        u32 const nrm = collect_tips(g, comp, rm);
        if (nrm == 0) break;
        for (u32 x = 0; x < nrm; ++x) remove_node(g, rm[x]);
      }
#ifdef MA_PROFILE
      g.dbg_phase = phase;
#endif
      compress_graph(g, comp);
      if (phase == 0) CPROF_ACC(3);
    }
    CPROF_ACC(6);
    if (g.flags & 4u) break;

    // ---- BuildTraversalIndex (traversal_index.cpp:34-119) ----
    u32* flat_of = g.scratch + NC;       // node -> flat
    u32* flat_nodes = g.scratch + 2 * NC;
    u32* rstart = g.scratch + 3 * NC;    // [2V]
    u32* rcnt = g.scratch + 5 * NC;      // [2V]
    u32 const EK = g.ek;
    u32* adj_state = g.scratch + 7 * NC;              // [E <= EK * NC]
    u32* adj_ord = g.scratch + (7 + EK) * NC;         // [E]
    u32* ord_src = g.scratch + (7 + 2 * EK) * NC;     // [E] source node of ordinal
    u32* ord_val = g.scratch + (7 + 3 * EK) * NC;     // [E] dst<<2|kind of ordinal
    u8* traversed = reinterpret_cast<u8*>(g.scratch + (7 + 4 * EK) * NC);  // [E]
    u8* color = reinterpret_cast<u8*>(g.scratch + (8 + 4 * EK) * NC);      // [2V]
    u32* stack = g.scratch + (9 + 4 * EK) * NC;       // DFS frames (2 u32 each) / walk pool afterwards
    u32 V = 0, E = 0;
    for (u32 base = 0; base < g.n; base += 64) {
      u32 const i = base + lane;
      bool const in = i < g.n && g.alive[i] && g.comp[i] == comp;
      unsigned long long const m = __ballot(in);
      if (i < g.n) flat_of[i] = in ? V + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull))) : kNoNode;
      if (in) flat_nodes[V + __popcll(m & ((1ull << lane) - 1ull))] = i;
      V += static_cast<u32>(__popcll(m));
    }
    wave_sync_mem(g.lds);
    // one lane per node: edge counts per strand side, then the two prefix sums of the reference's loops in one --
    // a state's block starts where the edges of the nodes before it end, and so do the node's edge ordinals
    {
      u32 carry = 0;
      for (u32 f0 = 0; f0 < V; f0 += 64) {
        u32 const f = f0 + lane;
        u32 c0 = 0, c1 = 0;
        if (f < V) {
          u32 const i = flat_nodes[f];
          u32 const ne = g.nedge[i];
          for (u32 x = 0; x < ne; ++x) {
            u32 const e = g.edge[i * g.ecap + x];
            if (flat_of[e >> 2] == kNoNode) continue;
            if ((e >> 1) & 1u) c1++; else c0++;
          }
        }
        u32 incl = c0 + c1;
#pragma unroll
        for (u32 o = 1; o < 64; o <<= 1) {
          u32 const y = __shfl_up(incl, o, 64);
          if (lane >= o) incl += y;
        }
        if (f < V) {
          u32 const base = carry + incl - (c0 + c1);
          rstart[f * 2] = base;
          rstart[f * 2 + 1] = base + c0;
          rcnt[f * 2] = c0;
          rcnt[f * 2 + 1] = c1;
        }
        carry += __shfl(incl, 63, 64);
      }
      E = carry;
      wave_sync_mem(g.lds);
    }
    if (E > EK * NC || 2 * V > 2 * NC || 4u * V + 4u > g.wk * NC) {
      g.flags |= 4u;
      g.dbg_why = __LINE__;
      break;
    }
    for (u32 f = lane; f < V; f += 64) {
      u32 const i = flat_nodes[f];
      u32 const ne = g.nedge[i];
      u32 const b0 = rstart[f * 2], b1 = rstart[f * 2 + 1];
      u32 k0 = 0, k1 = 0;
      for (u32 x = 0; x < ne; ++x) {
        u32 const e = g.edge[i * g.ecap + x];
        u32 const df = flat_of[e >> 2];
        if (df == kNoNode) continue;
        u32 const ord = b0 + k0 + k1;  // ordinals count the node's edges in list order, whatever their side
        u32 const slot = ((e >> 1) & 1u) ? b1 + k1++ : b0 + k0++;
        ord_src[ord] = i;
        ord_val[ord] = e;
        adj_state[slot] = df * 2 + (e & 1u);
        adj_ord[slot] = ord;
      }
    }
    wave_sync_mem(g.lds);
    u32 const src_state = flat_of[g.source] * 2 + (g.sign[g.source] ? 0u : 1u);
    u32 const snk_flat = flat_of[g.sink];

    CPROF_ACC(7);
    // ---- HasCycle (cycle_finder.cpp:55-100) ----
    {
      for (u32 s = 0; s < 2 * V; ++s) color[s] = 0;
      u32 sp = 0;
      color[src_state] = 1;
      stack[0] = src_state;
      stack[1] = 0;
      sp = 1;
      bool cyc = false;
      while (sp > 0 && !cyc) {
        u32 const st = stack[(sp - 1) * 2], pos = stack[(sp - 1) * 2 + 1];
        if (pos >= rcnt[st]) {
          color[st] = 2;
          sp--;
          continue;
        }
        u32 const dsts = adj_state[rstart[st] + pos];
        stack[(sp - 1) * 2 + 1] = pos + 1;
        if (color[dsts] == 1) {
          cyc = true;
          break;
        }
        if (color[dsts] != 0) continue;
        color[dsts] = 1;
        stack[sp * 2] = dsts;
        stack[sp * 2 + 1] = 0;
        sp++;
      }
      if (cyc) {
        retry = true;
        break;
      }
    }

    // ---- ComputeGraphComplexity (graph_complexity.cpp:16-93) ----
    u32 cx_cc = 0, cx_bp = 0, cx_maxdeg = 0;
    f64 cx_unitig = 0.0, cx_cv = 0.0, cx_tip = 0.0;
    {
      u32 nn = V, ne = 0, unitigs = 0;
      OnlineStats cov, tip, uni;
      // degrees and coverages one lane per node; the f64 running statistics then take the nodes in index order
      // from registers (Welford's update is order-sensitive in its last bits)
      u32 ne_l = 0, mx_l = 0, bp_l = 0, un_l = 0;
      for (u32 f0 = 0; f0 < V; f0 += 64) {
        u32 const f = f0 + lane;
        u32 tot = 0, cls = 0;  // cls: 1 = tip, 2 = unitig, 0 = neither
        if (f < V) {
          u32 const i = flat_nodes[f];
          u32 d = 0, o = 0;
          u32 const self_minus = g.sign[i] ? 0u : 1u;
          u32 const nei = g.nedge[i];
          for (u32 x = 0; x < nei; ++x) {
            if (((g.edge[i * g.ecap + x] >> 1) & 1u) == self_minus) d++; else o++;
          }
          ne_l += d + o;
          mx_l = max(mx_l, max(d, o));
          if (d >= 2 || o >= 2) bp_l++;
          if (d == 1 && o == 1) un_l++;
          tot = nd_total(g, i);
          cls = (d == 0 || o == 0) ? 1u : ((d == 1 && o == 1) ? 2u : 0u);
        }
        u32 const cntb = V - f0 < 64u ? V - f0 : 64u;
        for (u32 l = 0; l < cntb; ++l) {
          f64 const cv = static_cast<f64>(__builtin_amdgcn_readlane(tot, l));
          u32 const c = __builtin_amdgcn_readlane(cls, l);
          cov.add(cv);
          if (c == 1u) tip.add(cv); else if (c == 2u) uni.add(cv);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ne_l += static_cast<u32>(__shfl_xor(ne_l, o));
        bp_l += static_cast<u32>(__shfl_xor(bp_l, o));
        un_l += static_cast<u32>(__shfl_xor(un_l, o));
        mx_l = max(mx_l, static_cast<u32>(__shfl_xor(mx_l, o)));
      }
      ne = ne_l;
      cx_bp = bp_l;
      unitigs = un_l;
      cx_maxdeg = mx_l;
      ne /= 2;
      cx_cc = ne >= nn ? ne - nn + 1 : 0;
      cx_unitig = nn > 0 ? static_cast<f64>(unitigs) / static_cast<f64>(nn) : 0.0;
      if (cov.n > 0 && cov.m1 > 0.0) cx_cv = cov.sd() / cov.m1;
      if (tip.n > 0 && uni.n > 0 && uni.m1 > 0.0) cx_tip = tip.m1 / uni.m1;
    }
    if (cx_cc >= 50 && cx_bp >= 50) {  // GraphComplexity::IsComplex (graph_complexity.h:112-121)
      retry = true;
      break;
    }

    CPROF_ACC(8);
    // ---- BuildHaplotypes: MaxFlow::NextPath loop (max_flow.cpp:162-280, graph.cpp:846-891) ----
    for (u32 e = 0; e < E; ++e) traversed[e] = 0;
    uint4* const arena = g.arena;
    u32 const arena_cap = g.ac;
    u32* walk_pool = stack;  // ordinals of accepted walks, back to back
    u32 walk_pool_cap = g.wk * NC, walk_pool_used = 0;
    u32* const walk_off = sh.walk_off;
    u32* const walk_len = sh.walk_len;
    u32* const walk_minw = sh.walk_minw;
    int nwalks = 0;
    bool hit_limit = false, arena_over = false;
    // The reference's breadth-first search enumerates walk PREFIXES: with b bubbles between source and sink that is
    // 2^b queue entries.  Entries of one level that stand on the same state expand identically, in the same relative
    // order, so only the earliest of them can be an ancestor of the first qualifying arrival at the sink -- the
    // earliest overall and, if that one has not crossed a new edge yet, the earliest that has.  The others are
    // folded into those two as a multiplicity, which is all that is needed to know how many entries the reference
    // would have popped (its 2^20-visit cap, max_flow.h:69).  Same walks, arena use linear in the graph.
    // rep [state][2] (arena index of the representatives in the level being built) + Confidence of every node: in the LDS
    // table when the component fits it, else behind the walk pool (deep panels keep components of many hundred nodes
    // after pruning: searched UNFOLDED they ran into the reference's 2^20-pop cap one HBM round trip at a time -- seconds
    // per window).  The last 4 V words of the pool are the work area of arrival_rank.
    bool const fold_lds = 5u * V <= g.link_cap;
    bool const fold = fold_lds || 9u * V + 64u <= walk_pool_cap;
    u32* const ftab = fold_lds ? l_link : walk_pool + (walk_pool_cap - 9u * V);
    if (fold && !fold_lds) walk_pool_cap -= 9u * V;
    u32* rep = ftab;
    // Node::Confidence (f64 arithmetic) of every node of the component, once: the search asks for it per outgoing
    // edge of every popped entry
    if (fold) {
      for (u32 f = lane; f < V; f += 64) ftab[4u * V + f] = nd_confidence(g, flat_nodes[f]);
      wave_sync_mem(g.lds);
      // EnqueueOutgoingEdges sorts a state's edges by the destination's Confidence (stable, descending) every time the
      // state is popped; the confidences do not change during the search, so each state's block is sorted ONCE, a lane
      // per state (the cycle check and the metrics above have read it in list order already)
      for (u32 stt = lane; stt < 2u * V; stt += 64) {
        u32 const c = rcnt[stt], b0 = rstart[stt];
        for (u32 x = 1; x < c; ++x) {
          u32 const as = adj_state[b0 + x], ao = adj_ord[b0 + x], cf = ftab[4u * V + (as >> 1)];
          u32 j = x;
          while (j > 0 && ftab[4u * V + (adj_state[b0 + j - 1] >> 1)] < cf) {
            adj_state[b0 + j] = adj_state[b0 + j - 1];
            adj_ord[b0 + j] = adj_ord[b0 + j - 1];
            --j;
          }
          adj_state[b0 + j] = as;
          adj_ord[b0 + j] = ao;
        }
      }
      wave_sync_mem(g.lds);
    }
    // Small graphs (every pruned component of the bench): (state, ordinal) of an edge in one word, (start, count) of a
    // state in one word, the traversed flags as bits in registers -- a popped entry then costs a handful of dependent
    // look-ups instead of some twenty-five
    // Where the packed tables live: the LDS kernels pack them in place (their scratch IS LDS); the HBM kernels copy them
    // behind the fold tables when the component fits the LDS table, so that the search itself never leaves LDS (the
    // arena only takes fire-and-forget stores and one lane-parallel read per level).  Otherwise: the plain search.
    u32 const tab_words = 2u * V + E + (E + 31u) / 32u;
    bool const tabs_in_link = !g.lds && fold_lds && 5u * V + tab_words <= g.link_cap;
    bool const fastq = fold && (g.lds || tabs_in_link) && E < 0x10000u && 2u * V < 0x10000u && (!g.lds || (E + 31u) / 32u <= NC);
    u32* const rs_tab = tabs_in_link ? l_link + 5u * V : rcnt;            // [2V] start | count << 16
    u32* const adjp = tabs_in_link ? l_link + 7u * V : adj_state;         // [E]  state | ordinal << 16
    u32* const trav_w = tabs_in_link ? l_link + 7u * V + E : g.scratch;   // [E / 32] traversed flags (the list area is free here)
    if (fastq) {
      for (u32 x = lane; x < E; x += 64) adjp[x] = adj_state[x] | (adj_ord[x] << 16);
      for (u32 stt = lane; stt < 2u * V; stt += 64) rs_tab[stt] = rstart[stt] | (rcnt[stt] << 16);
      for (u32 x = lane; x < (E + 31u) / 32u; x += 64) trav_w[x] = 0;
      wave_sync_mem(g.lds);
    }
    while (true) {
      u32 an = 0, head = 0;
      u32 next_begin = 0;          // first arena index of the level being built
      u64 build_total = 0;         // reference entries (multiplicities) of the level being built
      if (fold) {
        for (u32 x = lane; x < 4u * V; x += 64) rep[x] = kNoNode;
        wave_sync_mem(g.lds);
      }
      // w of an arena record: bit 31 = the walk has crossed a not yet traversed edge, bits 0-30 = multiplicity
      auto push = [&](u32 ord, u32 st, u32 parent, u32 flag, u32 mult) {
        build_total += mult;
        if (fold) {
          u32 const r0 = rep[st * 2];
          if (r0 != kNoNode && r0 >= next_begin) {
            u32 const w0 = arena[r0].w;
            if ((w0 >> 31) || !flag) {
              arena[r0].w = (w0 & 0x80000000u) | min((w0 & 0x7FFFFFFFu) + mult, 0x40000000u);
              return;
            }
            u32 const r1 = rep[st * 2 + 1];
            if (r1 != kNoNode && r1 >= next_begin) {
              arena[r1].w = 0x80000000u | min((arena[r1].w & 0x7FFFFFFFu) + mult, 0x40000000u);
              return;
            }
            if (an >= arena_cap) {
              arena_over = true;
              return;
            }
            rep[st * 2 + 1] = an;
            arena[an++] = make_uint4(ord, st, parent, 0x80000000u | mult);
            return;
          }
          if (an < arena_cap) rep[st * 2] = an;
        }
        if (an >= arena_cap) {
          arena_over = true;
          return;
        }
        arena[an++] = make_uint4(ord, st, parent, (flag << 31) | mult);
      };
      // EnqueueOutgoingEdges (max_flow.cpp:235-280): stable sort by dst Confidence desc, new edges first
      auto enqueue = [&](u32 state, u32 parent, u32 pw) {
        u32 const cnt = rcnt[state];
        if (cnt == 0) return;
        u32* const idx = sh.eq_idx;
        u32* const conf = sh.eq_conf;
        u32 const m = cnt < static_cast<u32>(kEdgeCap) ? cnt : static_cast<u32>(kEdgeCap);
        for (u32 x = 0; x < m; ++x) {
          u32 const p = rstart[state] + x;
          u32 const cf = fold ? ftab[4u * V + (adj_state[p] >> 1)] : nd_confidence(g, flat_nodes[adj_state[p] >> 1]);
          u32 j = x;
          while (j > 0 && conf[j - 1] < cf) {
            conf[j] = conf[j - 1];
            idx[j] = idx[j - 1];
            --j;
          }
          conf[j] = cf;
          idx[j] = p;
        }
        for (int pass = 0; pass < 2; ++pass)
          for (u32 x = 0; x < m; ++x) {
            u32 const p = idx[x];
            bool const trav = traversed[adj_ord[p]] != 0;
            if (trav != (pass == 1)) continue;
            push(adj_ord[p], adj_state[p], parent, (pw >> 31) | (trav ? 0u : 1u), pw & 0x7FFFFFFFu);
            if (arena_over) return;
          }
      };
      auto is_trav = [&](u32 ord) { return (trav_w[ord >> 5] >> (ord & 31u)) & 1u; };
      auto enqueue_fast = [&](u32 state, u32 parent, u32 pw) {
        u32 const rs = rs_tab[state], cnt = rs >> 16, b0 = rs & 0xFFFFu;
        if (cnt == 0) return;
        u32 const pflag = pw >> 31, mult = pw & 0x7FFFFFFFu;
        if (cnt == 1) {
          u32 const ap = adjp[b0], ord = ap >> 16;
          push(ord, ap & 0xFFFFu, parent, pflag | (is_trav(ord) ? 0u : 1u), mult);
          return;
        }
        u32 const m = cnt < 8u ? cnt : 8u;  // (a state of the bidirected graph has at most four edges)
        u32 aps[8];
#pragma unroll
        for (u32 x = 0; x < 8; ++x) aps[x] = adjp[b0 + (x < m ? x : 0u)];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
          for (u32 x = 0; x < 8; ++x) {
            if (x >= m) continue;
            u32 const ord = aps[x] >> 16;
            bool const trav = is_trav(ord) != 0;
            if (trav != (pass == 1)) continue;
            push(ord, aps[x] & 0xFFFFu, parent, pflag | (trav ? 0u : 1u), mult);
            if (arena_over) return;
          }
        }
      };
      if (fastq) enqueue_fast(src_state, kNoParent, 1u); else enqueue(src_state, kNoParent, 1u);
      u64 pops_before = 0, lvl_total = build_total;  // reference pops before / inside the level being popped
      u32 lvl_end = an;
      build_total = 0;
      next_begin = an;
      bool undecided = false;
      u64 const limit = static_cast<u64>(P.bfs_limit);
      i64 best = -1;
      bool straddle = false;  // the cap falls inside the level being popped, which holds a qualifying arrival
      auto enter_level = [&]() {  // the reference stops at its (limit + 1)-th pop: does that fall into this level?
        if (pops_before + lvl_total <= limit) return;
        if (pops_before < limit) {
          // it falls inside; if no entry of the level qualifies the reference pops them all and gives up, otherwise
          // the question is whether the first qualifying arrival comes before the cap: its exact position in the
          // reference's (unfolded) queue is worked out once it is found (arrival_rank below)
          for (u32 x = head; x < lvl_end; ++x) {
            uint4 const e = arena[x];
            if ((e.y >> 1) == snk_flat && (e.w >> 31)) straddle = true;
          }
        }
        if (!straddle) hit_limit = true;
      };
      enter_level();
      while (head < an && !arena_over && !hit_limit && !undecided) {
        if (head == lvl_end) {
          if (straddle) break;  // (the arrival the level promised was not found: cannot happen)
          pops_before += lvl_total;
          lvl_total = build_total;
          build_total = 0;
          lvl_end = an;
          next_begin = an;
          enter_level();
          if (hit_limit || undecided) break;
        }
        if (fastq && lvl_end - head <= 64u) {
          // The whole level at once: a lane per entry fetches the entry, its state's block and (up to) the two edges most
          // states have -- three dependent look-ups per LEVEL -- and the entries are then popped in order out of
          // registers; what stays serial per popped entry is the fold bookkeeping of push().
          u32 const nlev = lvl_end - head;
          u32 my_st = 0, my_w = 0, my_cnt = 0, my_a0 = 0, my_a1 = 0, my_b0 = 0, my_t = 0;
          if (lane < nlev) {
            uint4 const e = arena[head + lane];
            my_st = e.y;
            my_w = e.w;
            u32 const rs = rs_tab[e.y];
            my_cnt = rs >> 16;
            my_b0 = rs & 0xFFFFu;
            my_a0 = adjp[my_b0];                         // (entry 0 of a block that may be empty: any word of the table)
            my_a1 = adjp[my_b0 + (my_cnt > 1u ? 1u : 0u)];
            my_t = (my_cnt >= 1u ? is_trav(my_a0 >> 16) : 0u) | ((my_cnt >= 2u ? is_trav(my_a1 >> 16) : 0u) << 1);
          }
          bool stop = false;
          for (u32 x = 0; x < nlev; ++x) {
            u32 const ai = head + x;
            u32 const y = __builtin_amdgcn_readlane(my_st, x), ww = __builtin_amdgcn_readlane(my_w, x);
            if ((y >> 1) == snk_flat) {
              if (!(ww >> 31)) continue;
              best = ai;
              stop = true;
              break;
            }
            u32 const cnt = __builtin_amdgcn_readlane(my_cnt, x);
            if (cnt == 0) continue;
            u32 const pflag = ww >> 31, mult = ww & 0x7FFFFFFFu;
            if (cnt <= 2u) {
              u32 const a0 = __builtin_amdgcn_readlane(my_a0, x), a1 = __builtin_amdgcn_readlane(my_a1, x);
              u32 const tb = __builtin_amdgcn_readlane(my_t, x);
              bool const t0 = (tb & 1u) != 0;
              if (cnt == 1u) {
                push(a0 >> 16, a0 & 0xFFFFu, ai, pflag | (t0 ? 0u : 1u), mult);
              } else {
                bool const t1 = (tb & 2u) != 0;
                // new edges first, each class in confidence order: only (old, new) swaps the two
                u32 const f0 = (t0 && !t1) ? a1 : a0, f1 = (t0 && !t1) ? a0 : a1;
                bool const ft0 = (t0 && !t1) ? t1 : t0, ft1 = (t0 && !t1) ? t0 : t1;
                push(f0 >> 16, f0 & 0xFFFFu, ai, pflag | (ft0 ? 0u : 1u), mult);
                if (!arena_over) push(f1 >> 16, f1 & 0xFFFFu, ai, pflag | (ft1 ? 0u : 1u), mult);
              }
            } else {
              enqueue_fast(y, ai, ww);
            }
            if (arena_over) {
              stop = true;
              break;
            }
          }
          head = lvl_end;
          if (stop) break;
          continue;
        }
        u32 const ai = head++;
        uint4 const wn = arena[ai];
        if ((wn.y >> 1) == snk_flat) {
          if (!(wn.w >> 31)) continue;
          best = ai;
          break;
        }
        if (fastq) enqueue_fast(wn.y, ai, wn.w); else enqueue(wn.y, ai, wn.w);
      }
      if (straddle && best < 0) undecided = true;
      if (straddle && best >= 0) {
        // Position of the arrival inside its level of the REFERENCE's queue.  Entries of a level come in the order of their
        // parents, children of one parent in enqueue order; the number of entries that precede the arrival's ancestor A_i
        // in level i, per state, is B_i[s'] = sum over s of B_(i-1)[s] x edges(s -> s')  (every entry before A_(i-1)
        // expands into all its edges; sink entries into none)  +  the siblings enqueued before A_i.  The arrival is the
        // (sum of B_d)-th entry of its level; the reference reaches it iff that is below the pops the cap leaves.
        u32 depth = 0;
        for (u32 i = static_cast<u32>(best); i != kNoParent; i = arena[i].z) depth++;
        if (walk_pool_used + depth + (fold_lds ? 4u * V : 0u) > walk_pool_cap) {
          undecided = true;
        } else {
          u32* const chain = walk_pool + walk_pool_used;  // arena indices, level 0 first
          {
            u32 pos = depth;
            for (u32 i = static_cast<u32>(best); i != kNoParent; i = arena[i].z) chain[--pos] = i;
          }
          u32* bcur = fold_lds ? walk_pool + walk_pool_cap - 4u * V : ftab + 5u * V;
          u32* bnxt = bcur + 2u * V;
          constexpr u32 kSat = 1u << 22;
          auto block_of = [&](u32 stt, u32* b0, u32* cnt) {
            if (fastq) {
              u32 const rs = rs_tab[stt];
              *b0 = rs & 0xFFFFu;
              *cnt = rs >> 16;
            } else {
              *b0 = rstart[stt];
              *cnt = rcnt[stt];
            }
          };
          auto edge_of = [&](u32 p, u32* dst, u32* ord) {
            if (fastq) {
              u32 const ap = adjp[p];
              *dst = ap & 0xFFFFu;
              *ord = ap >> 16;
            } else {
              *dst = adj_state[p];
              *ord = adj_ord[p];
            }
          };
          auto trav_of = [&](u32 ord) { return fastq ? (is_trav(ord) != 0) : (traversed[ord] != 0); };
          for (u32 x = lane; x < 2u * V; x += 64) bcur[x] = 0;
          wave_sync_mem(g.lds);
          u32 parent_state = src_state;
          for (u32 lv = 0; lv < depth; ++lv) {
            uint4 const A_ = arena[chain[lv]];
            if (lv > 0) {
              for (u32 x = lane; x < 2u * V; x += 64) bnxt[x] = 0;
              wave_sync_mem(g.lds);
              for (u32 stt = lane; stt < 2u * V; stt += 64) {
                u32 const c = bcur[stt];
                if (c == 0 || (stt >> 1) == snk_flat) continue;
                u32 b0, cnt;
                block_of(stt, &b0, &cnt);
                for (u32 x = 0; x < cnt; ++x) {
                  u32 dst, ord;
                  edge_of(b0 + x, &dst, &ord);
                  atomicAdd(&bnxt[dst], c);
                }
              }
              wave_sync_mem(g.lds);
              for (u32 x = lane; x < 2u * V; x += 64) bnxt[x] = min(bnxt[x], kSat);
              wave_sync_mem(g.lds);
              u32* const tsw = bcur;
              bcur = bnxt;
              bnxt = tsw;
            }
            // siblings enqueued before A_: new edges first, each class in confidence order (the blocks are sorted)
            {
              u32 b0, cnt;
              block_of(parent_state, &b0, &cnt);
              bool found = false;
              for (int pass = 0; pass < 2 && !found; ++pass)
                for (u32 x = 0; x < cnt && !found; ++x) {
                  u32 dst, ord;
                  edge_of(b0 + x, &dst, &ord);
                  if (trav_of(ord) != (pass == 1)) continue;
                  if (ord == A_.x) found = true;
                  else if (lane == 0) bcur[dst] = min(bcur[dst] + 1u, kSat);
                }
              wave_sync_mem(g.lds);
            }
            parent_state = A_.y;
          }
          u64 rank = 0;
          for (u32 x0 = 0; x0 < 2u * V; x0 += 64) {
            u32 vv = x0 + lane < 2u * V ? bcur[x0 + lane] : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) vv += static_cast<u32>(__shfl_xor(vv, o));
            rank += vv;
          }
          if (pops_before + rank >= limit) {  // the reference's (limit + 1)-th pop comes first: it gives up
            best = -1;
            hit_limit = true;
          }
        }
      }
      if (undecided) arena_over = true;  // reported as a capacity failure below
      if (best < 0) break;
      // reconstruct (max_flow.cpp:42-54) into the pool, reversed to source->sink order
      u32 wl = 0;
      for (u32 i = static_cast<u32>(best); i != kNoParent; i = arena[i].z) wl++;
      if (compact && nwalks < kMaxWalks && walk_pool_used + wl > walk_pool_cap) {
        g.flags |= 4u;  // the compact graph's smaller pool: k_clean has the window again
        g.dbg_why = __LINE__;
        break;
      }
      if (nwalks >= kMaxWalks || walk_pool_used + wl > walk_pool_cap) {
        status |= MA_W_HAP_OVERFLOW;
        break;
      }
      u32 const off = walk_pool_used;
      {
        u32 pos = wl;
        for (u32 i = static_cast<u32>(best); i != kNoParent; i = arena[i].z) {
          u32 const ord = arena[i].x;
          walk_pool[off + --pos] = ord;
          if (fastq) {
            if (lane == 0) trav_w[ord >> 5] |= 1u << (ord & 31u);
          } else {
            traversed[ord] = 1;
          }
        }
      }
      wave_sync_mem(g.lds);
      // MinWeight over the walk's nodes (path.cpp:34-37): a lane per edge, the source node on lane 0 as well
      auto conf_of = [&](u32 node) { return fold ? ftab[4u * V + flat_of[node]] : nd_confidence(g, node); };
      u32 mw = 0xFFFFFFFFu;
      for (u32 x = lane; x < wl; x += 64) {
        u32 const ordx = walk_pool[off + x];
        mw = min(mw, conf_of(ord_val[ordx] >> 2));
        if (x == 0) mw = min(mw, conf_of(ord_src[ordx]));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mw = min(mw, static_cast<u32>(__shfl_xor(mw, o)));
      walk_off[nwalks] = off;
      walk_len[nwalks] = wl;
      walk_minw[nwalks] = mw;
      nwalks++;
      walk_pool_used += wl;
    }
    CPROF_ACC(9);
    if (compact && arena_over) {
      g.flags |= 4u;
      g.dbg_why = __LINE__;
    }
    if (g.flags & 4u) break;
    if (arena_over) status |= MA_W_TABLE_OVERFLOW;
    if (arena_over && nwalks == 0) {
      // "no walk" cannot be told from "walk not reached before the arena filled up" (the reference allows 2^20
      // visits, max_flow.h:69): a capacity failure like the others -- flagged, and no attempt at the next k, which
      // would report haplotypes the reference never builds
      g.flags |= 4u;
      break;
    }
    if (hit_limit) status |= MA_W_BFS_LIMIT;
    if (nwalks == 0) continue;  // graph.cpp:225

    g.ranked = g.n <= g.link_cap;  // (distance, last slice) per slice
    CSUB_T0();
    if (g.ranked) rank_slices(g);
    // Node::Confidence and the total coverage of the component's nodes, behind the slice ranks: every walk asks for
    // both at every node it passes
    bool const have_tab = g.ranked && g.n + 2u * V <= g.link_cap;
    if (have_tab) {
      for (u32 f = lane; f < V; f += 64) {
        l_link[g.n + f] = nd_confidence(g, flat_nodes[f]);
        l_link[g.n + V + f] = nd_total(g, flat_nodes[f]);
      }
      wave_sync_mem(g.lds);
    }
    auto conf_at = [&](u32 node) { return have_tab ? l_link[g.n + flat_of[node]] : nd_confidence(g, node); };
    auto total_at = [&](u32 node) { return have_tab ? l_link[g.n + V + flat_of[node]] : nd_total(g, node); };
    // stable sort by MinWeight desc (graph.cpp:876-879)
    int* const order = sh.order;
    for (int i = 0; i < nwalks; ++i) {
      int j = i;
      while (j > 0 && walk_minw[order[j - 1]] < walk_minw[i]) {
        order[j] = order[j - 1];
        --j;
      }
      order[j] = i;
    }
    if (static_cast<int>(ncomp_out) >= MC || static_cast<int>(slot) + 1 > MH) {
      status |= MA_W_HAP_OVERFLOW;
      break;
    }
    size_t const cidx = static_cast<size_t>(w) * MC + ncomp_out;
    u32 const hap0 = slot;
    CSUB_ACC(11);
    // REF haplotype (graph.cpp:902-924)
    {
      u32* confs = g.scratch;  // reuse
      u32 ncf = 0;
      for (u32 f0 = 0; f0 < V; f0 += 64) {  // one lane per node, appended in node order
        u32 const f = f0 + lane;
        u32 const i = f < V ? flat_nodes[f] : 0u;
        bool const is_ref = f < V && (g.label[i] & 1u);
        unsigned long long const m = __ballot(is_ref);
        if (is_ref) confs[ncf + __popcll(m & ((1ull << lane) - 1ull))] = nd_confidence(g, i);
        ncf += static_cast<u32>(__popcll(m));
      }
      wave_sync_mem(g.lds);
      sort_small_u32(confs, ncf, g.lds);
      u32 const wgt = ncf == 0 ? 1u : median_sorted(confs, ncf);
      size_t const hi = static_cast<size_t>(w) * MH + slot;
      u32 len = anchor_len;
      if (len > static_cast<u32>(ML)) {
        status |= MA_W_LEN_OVERFLOW;
        len = ML;
      }
      for (u32 x0 = 0; x0 < len; x0 += 512) {  // eight loads in flight, then eight stores (the compiler cannot tell that
        u8 tmp[8];                              // the two buffers do not overlap and would wait for every byte)
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          u32 const x = x0 + q * 64 + lane;
          tmp[q] = x < len ? ref_anchor[x] : 0;
        }
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          u32 const x = x0 + q * 64 + lane;
          if (x < len) A.out.hap_bases[hi * ML + x] = tmp[q];
        }
      }
      wave_sync_mem(g.lds);
      A.out.hap_len[hi] = len;
      A.out.hap_nruns[hi] = 1;
      A.out.hap_runs[(hi * MR) * 2 + 0] = wgt;
      A.out.hap_runs[(hi * MR) * 2 + 1] = anchor_len;
      for (int x = 0; x < 6; ++x) A.out.hap_stats[hi * 6 + x] = 0.0;
      slot++;
    }
    CSUB_ACC(12);
    f64 max_alt_cv = -1.0;
    bool has_alt = false;
    for (int oi = 0; oi < nwalks; ++oi) {
      int const wi = order[oi];
      if (static_cast<int>(slot) >= MH) {
        status |= MA_W_HAP_OVERFLOW;
        break;
      }
      size_t const hi = static_cast<size_t>(w) * MH + slot;
      u8* hb = A.out.hap_bases + hi * ML;
      u32* covs = g.scratch;  // node coverages in walk order
      u32 ncov = 0, pos = 0, nruns = 0;
      bool ok = true, runs_ok = true;
      CSUB_ACC(15);
      // BuildSequence (max_flow.cpp:64-113)
      u32 const wl = walk_len[wi];
      const u32* wo = walk_pool + walk_off[wi];
      if (g.lds) {
        // The whole walk at once (every base string is in the LDS pool, stored orientation): a lane per node works out
        // where the node's bases go (prefix sum of the lengths each node contributes) and turns its slice list into
        // PIECES -- (first source byte, direction, complement, length, output offset) -- then the pieces are copied one
        // after the other, a lane per base.  Was: a call per node with its own scans, list look-up and fences.
        u32* const pieces = g.pieces;
        if (lane == 0) pieces[0] = 0;
        wave_sync_mem(true);
        u32 carry = 0;
        bool pover = false;
        for (u32 t0 = 0; t0 <= wl; t0 += 64) {
          u32 const tt = t0 + lane;
          bool const on = tt <= wl;
          u32 node = 0, clen = 0, skip = 0;
          bool dfl = true;
          if (on) {
            if (tt == 0) {
              u32 const e0 = ord_val[wo[0]];
              node = ord_src[wo[0]];
              dfl = ((e0 >> 1) & 1u) == 0u;  // walk[0].SrcSign() == PLUS
            } else {
              u32 const e = ord_val[wo[tt - 1]];
              node = e >> 2;
              dfl = (e & 1u) == 0u;          // conn.DstSign() == PLUS
              skip = K - 1;
            }
            u32 const ln = g.len[node];
            clen = ln > skip ? ln - skip : 0u;
          }
          u32 inc = clen;
#pragma unroll
          for (u32 o = 1; o < 64; o <<= 1) {
            u32 const y = __shfl_up(inc, o, 64);
            if (lane >= o) inc += y;
          }
          if (on) {
            covs[tt] = total_at(node);
            if (static_cast<int>(tt) < MR) {
              A.out.hap_runs[(hi * MR + tt) * 2 + 0] = conf_at(node);
              A.out.hap_runs[(hi * MR + tt) * 2 + 1] = clen;
            } else {
              runs_ok = false;
            }
            u32 o = carry + inc - clen;
            u32 sidx = dfl ? g.head[node] : g.tail[node];
            while (sidx != kNoNode) {
              u32 const d = g.sdesc[sidx], st = sd_st(d), ln = sd_ln(d), rc = sd_rc(d);
              u32 const bl = g.blen[sidx], po = g.bsrc[sidx] & 0x3FFFFFFFu;
              u32 const nx = dfl ? g.snext[sidx] : g.sprev[sidx];
              if (skip >= ln) {
                skip -= ln;
              } else {
                u32 const j0 = skip, take = ln - skip;
                skip = 0;
                // j-th base of the slice as the walk reads it: forwards  rc ? comp(B[bl-1-st-j]) : B[st+j],
                //                                             backwards rc ? B[bl-st-ln+j] : comp(B[st+ln-1-j])
                bool const back = dfl ? (rc != 0u) : (rc == 0u);  // source index runs downwards, bases complemented
                u32 const idx0 = dfl ? (rc ? bl - 1u - st - j0 : st + j0) : (rc ? bl - st - ln + j0 : st + ln - 1u - j0);
                u32 const at = atomicAdd(&pieces[0], 1u);
                if (at < g.piece_cap && o + take <= 0xFFFFu) {
                  pieces[1 + 2 * at] = (po + idx0) | (back ? 0x80000000u : 0u);
                  pieces[2 + 2 * at] = take | (o << 16);
                } else {
                  pover = true;
                }
                o += take;
              }
              sidx = nx;
            }
          }
          carry += __shfl(inc, 63, 64);
        }
        pos = carry;
        ncov = wl + 1u;
        nruns = wl + 1u;
        wave_sync_mem(true);
        if (__ballot(pover)) {
          g.flags |= 4u;  // more slices than the piece table holds: k_clean has the window again
          g.dbg_why = __LINE__;
          break;
        }
        u32 const np = pieces[0];
        const u8* const lp = g.pool;
        bool over = false;
        for (u32 q = 0; q < np; ++q) {
          u32 const w0 = pieces[1 + 2 * q], w1 = pieces[2 + 2 * q];
          u32 const src = w0 & 0x3FFFFFFFu, ln = w1 & 0xFFFFu, o = w1 >> 16;
          bool const back = (w0 >> 31) != 0u;
          for (u32 j = lane; j < ln; j += 64) {
            u8 const c = lp[back ? src - j : src + j];
            u32 const at = o + j;
            if (at < static_cast<u32>(ML)) hb[at] = back ? dev_complement(c) : c; else over = true;
          }
        }
        ok = __ballot(over) == 0;
      } else {
      {
        u32 const e0 = ord_val[wo[0]];
        u32 const sn = ord_src[wo[0]];
        bool const dflt = ((e0 >> 1) & 1u) == 0u;  // walk[0].SrcSign() == PLUS
        u32 const before = pos;
        ok &= emit_node_seq(g, sn, dflt, 0, hb, &pos, ML);
        covs[ncov++] = total_at(sn);
        if (static_cast<int>(nruns) < MR) {
          A.out.hap_runs[(hi * MR + nruns) * 2 + 0] = conf_at(sn);
          A.out.hap_runs[(hi * MR + nruns) * 2 + 1] = pos - before;
        } else runs_ok = false;
        nruns++;
      }
      for (u32 x = 0; x < wl; ++x) {
        u32 const e = ord_val[wo[x]];
        u32 const dn = e >> 2;
        bool const dflt = (e & 1u) == 0u;  // conn.DstSign() == PLUS
        u32 const before = pos;
        ok &= emit_node_seq(g, dn, dflt, K - 1, hb, &pos, ML);
        covs[ncov++] = total_at(dn);
        if (static_cast<int>(nruns) < MR) {
          A.out.hap_runs[(hi * MR + nruns) * 2 + 0] = conf_at(dn);
          A.out.hap_runs[(hi * MR + nruns) * 2 + 1] = pos - before;
        } else runs_ok = false;
        nruns++;
      }
      }
      CSUB_ACC(13);
      // dedup by sequence against kept haplotypes of this component and the ref anchor (graph.cpp:883-887)
      bool dup = false;
      if (ok) {
        for (u32 h = hap0; h < slot && !dup; ++h) {
          size_t const oi2 = static_cast<size_t>(w) * MH + h;
          if (A.out.hap_len[oi2] != pos) continue;
          const u8* ob = A.out.hap_bases + oi2 * ML;
          bool diff = false;
          for (u32 x = lane; x < pos; x += 64) diff |= ob[x] != hb[x];
          dup = __ballot(diff) == 0;
        }
      }
      if (dup) continue;
      if (!ok || !runs_ok) status |= MA_W_LEN_OVERFLOW;
      A.out.hap_len[hi] = pos < static_cast<u32>(ML) ? pos : static_cast<u32>(ML);
      A.out.hap_nruns[hi] = nruns < static_cast<u32>(MR) ? nruns : static_cast<u32>(MR);
      CSUB_ACC(14);
      // Path::Finalize (path.cpp:39-70)
      OnlineStats st;
      for (u32 x = 0; x < ncov; ++x) st.add(static_cast<f64>(covs[x]));
      f64 const mean = st.m1, sdv = st.sd();
      f64 const total = mean * static_cast<f64>(st.n);
      f64 cv = 0.0, qcv = 0.0;
      if (mean > 0.0) cv = sdv / mean;
      sort_small_u32(covs, ncov, g.lds);
      f64 const med = static_cast<f64>(median_sorted(covs, ncov));
      if (ncov >= 4) {
        f64 const q1 = static_cast<f64>(covs[ncov / 4]), q3 = static_cast<f64>(covs[(ncov * 3) / 4]);
        if (q3 + q1 > 0.0) qcv = (q3 - q1) / (q3 + q1);
      }
      f64* hs = A.out.hap_stats + hi * 6;
      hs[0] = mean; hs[1] = med; hs[2] = sdv; hs[3] = cv; hs[4] = qcv; hs[5] = total;
      max_alt_cv = has_alt ? (max_alt_cv > cv ? max_alt_cv : cv) : cv;
      has_alt = true;
      slot++;
    }
    CPROF_ACC(10);
    A.out.comp_anchor[cidx] = cand_soff[ci];
    A.out.comp_hap0[cidx] = hap0;
    A.out.comp_nhaps[cidx] = slot - hap0;
    A.out.comp_cx[cidx * 3 + 0] = cx_cc;
    A.out.comp_cx[cidx * 3 + 1] = cx_bp;
    A.out.comp_cx[cidx * 3 + 2] = cx_maxdeg;
    A.out.comp_cxf[cidx * 4 + 0] = cx_unitig;
    A.out.comp_cxf[cidx * 4 + 1] = cx_cv;
    A.out.comp_cxf[cidx * 4 + 2] = cx_tip;
    A.out.comp_cxf[cidx * 4 + 3] = has_alt ? max_alt_cv : -1.0;
    ncomp_out++;
  }

  if ((g.flags & 4u) && compact) {
    // a capacity of the compact route (edge slots, search arena, walk pool): nothing is reported from here, k_clean --
    // launched behind this kernel -- assembles the window from the raw graph, which nobody has touched
    if (lane == 0) {
      ws.cg_state[a] = 0u;
      ws.cg_hdr[static_cast<size_t>(a) * kCgHdr + 3] = 100000u + g.dbg_why;
    }
    return;
  }
  if (g.flags & 4u) {
    A.out.win_status[w] = MA_W_TABLE_OVERFLOW | MA_W_NO_HAPLOTYPE;
    A.out.win_ncomp[w] = 0;
    if (lane == 0) atomicOr(&ws.win_flags[w], 1u);
    return;
  }
  if (retry || ncomp_out == 0) {  // graph.cpp:230-234 / results.empty(): try the next k
    A.out.win_ncomp[w] = 0;
    A.out.win_status[w] = MA_W_NO_HAPLOTYPE;
    return;
  }
#ifdef MA_PROFILE
  if (w < 4096) {
    g_cwin[w * 4 + 0] = __builtin_amdgcn_s_memtime() - t_begin;
    g_cwin[w * 4 + 1] = g.n;
    if (w < 4096) for (int q = 0; q < 5; ++q) g_ctime[w * 6 + q] = g.dbg_ph[6 + q];
    if (w < 4096) g_ctime[w * 6 + 5] = slot;
    if (w < 4096) { g_cmerge[w*8+0]=g.dbg_merges[0]; g_cmerge[w*8+1]=g.dbg_merges[1]; g_cmerge[w*8+2]=g.dbg_maxwalk[0]; g_cmerge[w*8+3]=g.dbg_maxwalk[1]; g_cmerge[w*8+4]=g.dbg_walks[0]; g_cmerge[w*8+5]=g.dbg_walks[1]; g_cmerge[w*8+6]=g.n; }
    g_cwin[w * 4 + 2] = dbg_rounds;
    g_cwin[w * 4 + 3] = static_cast<unsigned long long>(ncand);
  }
#endif
  u32 nalt = 0;
  for (u32 c = 0; c < ncomp_out; ++c) nalt += A.out.comp_nhaps[static_cast<size_t>(w) * MC + c] - 1;
  if (nalt == 0) status |= MA_W_NO_HAPLOTYPE;  // variant_builder.cpp:231-240
  A.out.win_ncomp[w] = ncomp_out;
  A.out.win_status[w] = status;
  if (lane == 0) atomicOr(&ws.win_flags[w], 1u);
}

// The candidate loop on the COMPACT graph k_clean_chains left: a few dozen nodes whose base strings are original k-mers
// (nodes the first CompressGraph did not touch) or merged strings in the window's pool.  kLds: the whole graph, the
// traversal index, the search arena and every work list live in LDS (graphs of up to kTailV nodes: all but the deepest
// windows); otherwise the same code runs on the arrays in HBM.  Capacities of this route that a window outgrows (edge
// slots, arena, walk pool) hand it to k_clean, never to the caller.
// Two LDS images: most compact graphs have a few dozen nodes (bench: 40 on average, 95 at most) and fit the small one,
// of which seven share a CU; the large one takes four.  The kernel is latency bound with one wavefront per window, so the
// windows in flight per CU set its speed.
struct TailSmall {
  static constexpr u32 V = 64, Arena = 192, Link = 320, S = 2, Ek = 2, Wk = 5, Pieces = 127, Pool = 3072;
};
struct TailLarge {
  static constexpr u32 V = 128, Arena = 320, Link = 640, S = 4, Ek = 2, Wk = 5, Pieces = 255, Pool = 5120;
};
struct TailHbm {  // the arrays stay in HBM (components of up to a thousand nodes: deep panels); the LDS table is large enough
  // for the fold tables AND the packed search tables of such a component, so that the search itself runs out of LDS
  static constexpr u32 V = 1, Arena = 1, Link = 12288, S = 1, Ek = 1, Wk = 1, Pieces = 1, Pool = 4;
};
template <class C>
struct TailLdsT {
  u32 cnt[C::V * C::S], role[C::V * 2], bsrc[C::V], blen[C::V], len[C::V], comp[C::V];
  u32 edge[C::V * kCgEdgeCap];
  u32 head[C::V], tail[C::V], snext[C::V], sprev[C::V], sdesc[C::V];
  u8 label[C::V], sign[C::V], bsign[C::V], nedge[C::V], alive[C::V];
  u32 scratch[(9 + 4 * C::Ek + C::Wk) * C::V];
  uint4 arena[C::Arena];
  u32 pool[C::Pool / 4];
  u32 pieces[1 + 2 * C::Pieces];
};
template <class C>
__device__ __forceinline__ bool tail_fits(u32 V, u32 pool_used, u32 kk, int S) {
  return V <= C::V && pool_used + 4u + V * kk <= C::Pool && static_cast<u32>(S) <= C::S;
}

template <class C, bool kLds>
__global__ __launch_bounds__(64, 2) void k_clean_tail(CleanArgs A) {
  int const a = blockIdx.x;
  u32 const lane = threadIdx.x;
  GraphWs const& ws = A.ws;
  if (ws.cg_state[a] != 1u) return;
  const u32* hdr = ws.cg_hdr + static_cast<size_t>(a) * kCgHdr;
  u32 const V = hdr[0], pool_used = hdr[2];
  u32 const kk = static_cast<u32>(win_kmer(ws, static_cast<int>(ws.active[a])));
  // the smallest image that holds the window
  int const cls = tail_fits<TailSmall>(V, pool_used, kk, ws.num_samples) ? 0 : (tail_fits<TailLarge>(V, pool_used, kk, ws.num_samples) ? 1 : 2);
  constexpr int kMine = !kLds ? 2 : (C::V == TailSmall::V ? 0 : 1);
  if (cls != kMine) return;  // another instantiation's window
  int const w = static_cast<int>(ws.active[a]);
  size_t const nb = static_cast<size_t>(a) * ws.vc;
  __shared__ CleanLdsT<C::Link> sh;
  Win g;
  g.link = sh.link;
  g.ranked = false;
  g.refb = A.b.ref_bases + A.b.ref_off[w];
  g.readb = A.b.read_bases + A.b.read_off[A.b.read_win_off[w]];
  g.ref_len = A.b.ref_off[w + 1] - A.b.ref_off[w];
  g.k = win_kmer(ws, w);
  g.S = ws.num_samples;
  g.n = V;
  g.min_node_cov = A.prm.min_node_cov;
  g.min_anchor_cov = A.prm.min_anchor_cov;
  g.pool = ws.cg_pool + static_cast<size_t>(a) * ws.pool_cap;
  g.ecap = kCgEdgeCap;
  g.lds = kLds;
  g.source = g.sink = -1;
  g.flags = 0;
  g.dbg_why = 0;
  u32 NC;
  if constexpr (kLds) {
    __shared__ TailLdsT<C> tl;
    int const S = g.S;
    for (u32 i = lane; i < V * static_cast<u32>(S); i += 64) tl.cnt[i] = ws.cg_cnt[nb * S + i];
    for (u32 i = lane; i < V * 2u; i += 64) tl.role[i] = ws.cg_role[nb * 2 + i];
    for (u32 i = lane; i < V * kCgEdgeCap; i += 64) tl.edge[i] = ws.cg_edge[nb * kCgEdgeCap + i];
    for (u32 i = lane; i < V; i += 64) {
      tl.bsrc[i] = ws.cg_bsrc[nb + i];
      u32 const bl = ws.cg_blen[nb + i];
      tl.blen[i] = bl;
      tl.len[i] = ws.cg_len[nb + i];
      tl.comp[i] = ws.cg_comp[nb + i];
      tl.label[i] = ws.cg_label[nb + i];
      tl.sign[i] = ws.cg_sign[nb + i];
      tl.bsign[i] = ws.cg_bsign[nb + i];
      tl.nedge[i] = ws.cg_nedge[nb + i];
      tl.alive[i] = 1;
      tl.head[i] = tl.tail[i] = i;
      tl.snext[i] = tl.sprev[i] = kNoNode;
      tl.sdesc[i] = sd_make(0u, bl, 0u);
    }
    g.cnt = tl.cnt;
    g.role = tl.role;
    g.src = tl.bsrc;
    g.label = tl.label;
    g.sign = tl.sign;
    g.nedge = tl.nedge;
    g.edge = tl.edge;
    g.comp = tl.comp;
    g.len = tl.len;
    g.alive = tl.alive;
    g.head = tl.head;
    g.tail = tl.tail;
    g.snext = tl.snext;
    g.sprev = tl.sprev;
    g.sdesc = tl.sdesc;
    g.bsrc = tl.bsrc;
    g.blen = tl.blen;
    g.bsign = tl.bsign;
    // every base string into LDS: the merged strings as they are, the k-mers of untouched nodes in their stored
    // orientation -- spelling a haplotype then never waits for HBM (one round trip per slice, ~20 slices a walk)
    {
      u8* const lp = reinterpret_cast<u8*>(tl.pool);
      const u32* gp = reinterpret_cast<const u32*>(g.pool);
      for (u32 i = lane; i < (pool_used + 3u) / 4u; i += 64) tl.pool[i] = gp[i];
      u32 nsingle = 0;
      for (u32 i0 = 0; i0 < V; i0 += 64) {
        u32 const i = i0 + lane;
        u32 const sv = i < V ? tl.bsrc[i] : 0x40000000u;
        bool const single = !(sv & 0x40000000u);
        unsigned long long const m = __ballot(single);
        if (single) {
          u32 const off = ((pool_used + 3u) & ~3u) + (nsingle + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)))) * kk;
          const u8* p = (sv & 0x80000000u) ? g.readb + (sv & 0x3FFFFFFFu) : g.refb + sv;
          bool const plus = tl.bsign[i] != 0;
          for (u32 x = 0; x < kk; ++x) lp[off + x] = plus ? p[x] : dev_complement(p[kk - 1u - x]);
          tl.bsrc[i] = 0x40000000u | off;
          tl.bsign[i] = 1;
        }
        nsingle += static_cast<u32>(__popcll(m));
      }
      g.pool = lp;
    }
    g.scratch = tl.scratch;
    g.pieces = tl.pieces;
    g.piece_cap = C::Pieces;
    g.ek = C::Ek;
    g.wk = C::Wk;
    g.arena = tl.arena;
    g.ac = C::Arena;
    g.link_cap = C::Link;
    NC = C::V;
  } else {
    g.cnt = ws.cg_cnt + nb * g.S;
    g.role = ws.cg_role + nb * 2;
    g.src = ws.cg_bsrc + nb;
    g.label = ws.cg_label + nb;
    g.sign = ws.cg_sign + nb;
    g.nedge = ws.cg_nedge + nb;
    g.edge = ws.cg_edge + nb * kCgEdgeCap;
    g.comp = ws.cg_comp + nb;
    g.len = ws.cg_len + nb;
    g.alive = ws.cg_alive + nb;
    g.head = ws.cg_head + nb;
    g.tail = ws.cg_tail + nb;
    g.snext = ws.cg_snext + nb;
    g.sprev = ws.cg_sprev + nb;
    g.sdesc = ws.cg_sdesc + nb;
    g.bsrc = ws.cg_bsrc + nb;
    g.blen = ws.cg_blen + nb;
    g.bsign = ws.cg_bsign + nb;
    g.scratch = ws.cg_scratch + static_cast<size_t>(a) * ws.cg_sc * 32;
    g.pieces = nullptr;
    g.piece_cap = 0;
    g.ek = 4;
    g.wk = 7;
    g.arena = ws.arena + static_cast<size_t>(a) * ws.ac;
    g.ac = ws.ac;
    g.link_cap = C::Link;
    NC = ws.cg_sc;
    for (u32 i = lane; i < g.n; i += 64) {
      g.alive[i] = 1;
      g.head[i] = g.tail[i] = i;
      g.snext[i] = g.sprev[i] = kNoNode;
      g.sdesc[i] = sd_make(0u, g.blen[i], 0u);
    }
  }
  g.nc = NC;
#ifdef MA_PROFILE
  for (int q = 0; q < 6; ++q) g.dbg_t[q] = 0;
  for (int q = 0; q < 16; ++q) g.dbg_ph[q] = 0;
  g.dbg_phase = 0; g.dbg_merges[0] = g.dbg_merges[1] = g.dbg_maxwalk[0] = g.dbg_maxwalk[1] = g.dbg_walks[0] = g.dbg_walks[1] = 0;
#endif
  int const ncand = static_cast<int>(hdr[1]);
  if (lane < static_cast<u32>(ncand)) {
    const u32* c = hdr + 8 + 6 * lane;
    sh.cand_comp[lane] = c[0];
    sh.cand_size[lane] = c[1];
    sh.cand_src[lane] = c[2];
    sh.cand_snk[lane] = c[3];
    sh.cand_soff[lane] = c[4];
    sh.cand_koff[lane] = c[5];
  }
  wave_sync_mem(g.lds);
  clean_candidates(A, g, sh, a, w, ncand, 1, NC);
}

int run_clean_chains(ma_ctx* ctx, const DBatch& b, const GraphWs& ws, const ma_params_t& prm);  // chains.hip

int run_clean_pass(ma_ctx* ctx, const DBatch& b, const GraphWs& ws, const ma_asm_out_t& out) {
  if (ws.n_active == 0) return MA_OK;
  CleanArgs args{b, ws, out, ctx->prm};
  bool const use_chains = !getenv("MA_NO_CHAINS");  // (read per call: the tests switch routes inside one process)
  if (use_chains && ws.cg_state) {
    // the first CompressGraph of every candidate component as bulk-parallel LDS work, then the rest of the candidate loop
    // on the few dozen nodes it leaves; windows it does not take (graph beyond its capacities, unusual shapes) keep
    // cg_state = 0 and go through k_clean from the raw graph
    MA_HIP(ctx, hipMemsetAsync(ws.cg_state, 0, sizeof(u32) * static_cast<size_t>(ws.n_active), ctx->stream));
    MA_TRY_RC(run_clean_chains(ctx, b, ws, ctx->prm));
    ctx->tic("k_clean_tail");
    hipLaunchKernelGGL((k_clean_tail<TailSmall, true>), dim3(ws.n_active), dim3(64), 0, ctx->stream, args);
    ctx->toc();
    ctx->tic("k_clean_tail");
    hipLaunchKernelGGL((k_clean_tail<TailLarge, true>), dim3(ws.n_active), dim3(64), 0, ctx->stream, args);
    ctx->toc();
    ctx->tic("k_clean_tail");
    hipLaunchKernelGGL((k_clean_tail<TailHbm, false>), dim3(ws.n_active), dim3(64), 0, ctx->stream, args);
    ctx->toc();
  } else if (ws.cg_state) {
    MA_HIP(ctx, hipMemsetAsync(ws.cg_state, 0, sizeof(u32) * static_cast<size_t>(ws.n_active), ctx->stream));
  }
  if (use_chains && ws.cg_state && getenv("MA_VERBOSE")) {  // which windows fall through to k_clean, and how large they are
    std::vector<u32> st(ws.n_active), nn(ws.n_active), hd(static_cast<size_t>(ws.n_active) * kCgHdr);
    (void)hipMemcpyAsync(st.data(), ws.cg_state, 4 * st.size(), hipMemcpyDeviceToHost, ctx->stream);
    (void)hipMemcpyAsync(nn.data(), ws.n_nodes, 4 * nn.size(), hipMemcpyDeviceToHost, ctx->stream);
    (void)hipMemcpyAsync(hd.data(), ws.cg_hdr, 4 * hd.size(), hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    u32 fall = 0, fall_big = 0, nmax = 0, vmax = 0;
    std::string why;
    for (int i = 0; i < ws.n_active; ++i) {
      nmax = std::max(nmax, nn[i]);
      if (st[i] == 0) {
        fall++;
        fall_big += nn[i] > 4096u;
        if (fall <= 12) why += " " + std::to_string(hd[static_cast<size_t>(i) * kCgHdr + 3]);
      } else {
        vmax = std::max(vmax, hd[static_cast<size_t>(i) * kCgHdr]);
      }
    }
    fprintf(stderr, "[microasm] clean: %d attempts, %u left to k_clean (%u of them beyond 4096 nodes), largest graph %u nodes, largest compact graph %u; reasons (source lines; 100000+ = tail):%s\n",
            ws.n_active, fall, fall_big, nmax, vmax, why.c_str());
  }
  ctx->tic("k_clean");
  hipLaunchKernelGGL(k_clean, dim3(ws.n_active), dim3(64), 0, ctx->stream, args);
  ctx->toc();
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

}  // namespace ma
