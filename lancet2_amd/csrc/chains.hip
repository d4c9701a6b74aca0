// k_clean_chains: components, anchors and the FIRST CompressGraph of every candidate component (cbdg/graph.cpp:392-509,
// :558-645, the first line of PruneComponent :515-540) as bulk-parallel work on an LDS-resident graph -- one workgroup of
// 256 threads per window attempt, nothing chased through HBM.  What it leaves is the COMPACT graph (a few dozen nodes, merged
// strings in a byte pool) that k_clean_tail (clean.hip) carries through the rest of the candidate loop.
//
// Why the reference's order-dependent compaction parallelises (checked against the oracle's sequential CompressGraph on
// every config by tools/dbg/compress_model.cpp, the CPU model this kernel was transliterated from):
//   * On the raw graph every node is one k-mer.  Call a node FULLY PLAIN when it has two edges on opposite sides, no self
//     loop, is not an anchor, and both neighbours have at most two edges.  FindCompressibleEdge / IsPotentialBuddyEdge
//     (graph.cpp:688-799) only ever let fully plain nodes be absorbed, and only fully plain nodes or one-edge tip ends start
//     a walk.  Maximal runs of fully plain nodes -- SEGMENTS -- are paths c_0 .. c_{m-1} between two boundary nodes L, R
//     that no walk crosses; segments only interact through the ORDER of their boundaries' edge lists.
//   * Inside a segment the sequential loop (CompressGraph visits nodes in index order, CompressNode(dflt) then (!dflt))
//     is an interval process: merged blocks only ever sit at the two ends, and a turn by single c_j absorbs EVERYTHING on
//     its first side (if the neighbour on the other side is a unit, or a boundary that passes the buddy test -- only a
//     two-edge anchor does), then everything on the other side iff the boundary it has just reached is such a buddy.  A tip
//     end's turn swallows the whole segment.  Turns are taken in node-index order: argmin over the remaining singles.
//   * Only two things are serial: that argmin loop (a handful of turns per segment) and the floor-rounded running
//     averages of Node::Merge (node.cpp:81-112), one step per absorbed unit.  Both run per segment on a GROUP of 4 / 8 / 16
//     lanes (one lane per averaged value), all segments of the window side by side.
//   * Everything else is positional: a block's string is its owner's k-mer plus ONE base per absorbed k-mer, a nested
//     block contributes a substring (Kmer::Merge drops k-1 bases at the joint, kmer.cpp:48-109); every leaf byte finds its
//     place through one linear map per nesting level.  Edge CONTENT follows from the final adjacency, edge ORDER from the
//     time of each edge's last rewrite (EmplaceEdge appends, EraseEdge closes the gap: a rewritten edge moves to the end).
// Anything outside this picture (a node with more than four edges, a ring of plain nodes, more nodes than the LDS image
// holds, counts beyond 16 bits, ...) leaves cg_state = 0 and the window goes through k_clean from the raw graph.
#include <algorithm>

#include "graph_ws.h"

namespace ma {

namespace {

constexpr int kT = 256;          // threads per window
constexpr u32 kNone16 = 0xFFFFu;

// node flags (u16)
constexpr u32 F_NE = 7u;          // [2:0] edges: 0, 1, 2 or 3 = three to eight (the count is in the node's side-table entry)
constexpr u32 F_SIGN = 1u << 3;   // Kmer sign (1 = PLUS)
constexpr u32 F_LABEL_SH = 4;     // [6:4] label
constexpr u32 F_PLAIN = 1u << 7;
constexpr u32 F_INI = 1u << 8;    // fully plain
constexpr u32 F_SRC = 1u << 9;
constexpr u32 F_SNK = 1u << 10;
constexpr u32 F_SIDEL = 1u << 11; // side bit (src_minus) of the edge that faces the segment's L end
constexpr u32 F_CAND_SH = 12;     // [15:12] candidate index, 15 = none

struct ChainArgs {
  DBatch b;
  GraphWs ws;
  ma_params_t prm;
  u32 cap;    // nodes the LDS image holds
  u32 n_lo;   // windows with n_lo < n <= cap are this launch's
  u32 xw;     // u32 words per node of the X region (>= 2)
  u32 xe_cap;   // nodes with three or four edges
  u32 seg_cap;  // segments per window (even)
  u32 cl_cap;   // compact nodes the alive list holds (even)
  u32 stop;     // (developer A/B) leave after phase `stop`: the window then takes the raw route; 0 = run to the end
};

struct Lds {
  u32* e2;      // [cap] first two edges: e0 | e1 << 16 (three or four edges: e0 | index into xe << 16)
  u16* fl;      // [cap]
  u16* a1;      // [cap] component id, later position in segment + 1
  u32* x;       // [cap * xw] FastSV labels -> anchor offsets -> chain states -> averaged values (u16)
  u16* segi;    // [cap] segment of a fully plain node (and of a tip end that swallowed one)
  u16* p2n;     // [cap] segment position -> node
  u16* abs;     // [cap] who absorbed the node
  u16* cid;     // [cap] compact id
  u32* blk;     // [cap] block of an owner: (lo + 1) | (hi + 1) << 16
  u32* key;     // [cap] time of the last edge rewrite << 1 | slot
  u16* xe;      // [kXeCap * 8] edges 1 .. 7 of a node with three to eight edges, [7] = their number
  u16* seg_base;  // [kSegCap]
  u16* seg_m;
  u16* seg_l;
  u16* seg_r;
  u16* seg_fl;
  u16* cl;        // [cl_cap] compact id -> node, when the compact graph is that small
  u32* bmin;      // [cap / 32] minimum of each 32-entry block of p2n: node << 12 | index
  u32* cand;      // [6 * 16]
  u32* misc;      // [32] counters / flags / wave sums
};

// seg_fl bits
constexpr u32 SF_BL = 1u, SF_BR = 2u, SF_TIPL = 4u, SF_TIPR = 8u, SF_SLOTL_SH = 4, SF_SLOTR_SH = 6, SF_DML = 1u << 8, SF_DMR = 1u << 9;
constexpr u32 SF_DESC = 1u << 10;  // some node index is smaller than its left neighbour's (else: ascending along the segment)
// misc words
constexpr int M_PUNT = 0, M_XE = 1, M_NCOMP = 3, M_NCAND = 4, M_G = 5, M_V = 6, M_POOL = 7, M_WSUM = 8, M_CHANGED = 12;  // wsum: 4 words, changed: 2 words

__device__ __forceinline__ u32 kind_rev(u32 kind) { return (((kind & 1u) ^ 1u) << 1) | (((kind >> 1) & 1u) ^ 1u); }

__device__ __forceinline__ u32 edge_at(const Lds& L, u32 i, u32 x) {
  u32 const v = L.e2[i];
  if (x == 0) return v & 0xFFFFu;
  if ((L.fl[i] & F_NE) <= 2u) return v >> 16;
  return L.xe[(v >> 16) * 8u + (x - 1u)];
}
// number of edges of node i
__device__ __forceinline__ u32 node_ne(const Lds& L, u32 i) {
  u32 const c = L.fl[i] & F_NE;
  return c <= 2u ? c : L.xe[(L.e2[i] >> 16) * 8u + 7u];
}
// the edge of a PLAIN node on side s (src_minus == s)
__device__ __forceinline__ u32 side_edge(const Lds& L, u32 i, u32 s) {
  u32 const v = L.e2[i], e0 = v & 0xFFFFu;
  return ((e0 >> 1) & 1u) == s ? e0 : (v >> 16);
}
__device__ __forceinline__ u32 side_slot(const Lds& L, u32 i, u32 s) { return (((L.e2[i] & 0xFFFFu) >> 1) & 1u) == s ? 0u : 1u; }

// exclusive prefix sum over the 256 threads; *total = sum.  Two barriers.
__device__ __forceinline__ u32 block_excl_scan(const Lds& L, u32 v, u32* total) {
  u32 const lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (u32 o = 1; o < 64; o <<= 1) {
    u32 const y = __shfl_up(inc, o, 64);
    if (lane >= o) inc += y;
  }
  __syncthreads();  // (the wave sums of a previous scan have been read)
  if (lane == 63) L.misc[M_WSUM + wv] = inc;
  __syncthreads();
  u32 off = 0, tot = 0;
#pragma unroll
  for (u32 q = 0; q < 4; ++q) {
    u32 const sq = L.misc[M_WSUM + q];
    if (q < wv) off += sq;
    tot += sq;
  }
  *total = tot;
  return off + inc - v;
}

// floor(num / den) for num < 2^30, quotient < 2^17, den < 2^13: float estimate is within one of the quotient
__device__ __forceinline__ u32 div_floor(u32 num, u32 den) {
  float const r = __builtin_amdgcn_rcpf(static_cast<float>(den));
  u32 q = static_cast<u32>(static_cast<float>(num) * r);
  i32 const rem = static_cast<i32>(num - __umul24(q, den));
  if (rem < 0) --q;
  else if (static_cast<u32>(rem) >= den) ++q;
  return q;
}

__device__ __forceinline__ u32 top_owner(const Lds& L, u32 q) {
  while (L.abs[q] != kNone16) q = L.abs[q];
  return q;
}

#ifdef MA_PROFILE
__device__ unsigned long long g_chprof[16];
#define CH_T0() unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define CH_ACC(slot)                                                      \
  do {                                                                    \
    unsigned long long const _t1 = __builtin_amdgcn_s_memtime();          \
    if (threadIdx.x == 0) atomicAdd(&g_chprof[slot], _t1 - _t0);          \
    _t0 = _t1;                                                            \
    if (A.stop == (slot) + 1u) return;                                    \
  } while (0)
#define CH_SUB0() unsigned long long _s0 = __builtin_amdgcn_s_memtime()
#define CH_SUB(slot)                                                      \
  do {                                                                    \
    unsigned long long const _s1 = __builtin_amdgcn_s_memtime();          \
    if (threadIdx.x == 0) atomicAdd(&g_chprof[slot], _s1 - _s0);          \
    _s0 = _s1;                                                            \
  } while (0)
#else
#define CH_T0() do {} while (0)
#define CH_ACC(slot) do { if (A.stop == (slot) + 1u) return; } while (0)
#define CH_SUB0() do {} while (0)
#define CH_SUB(slot) do {} while (0)
#endif

}  // namespace

#ifdef MA_PROFILE
extern "C" void ma_debug_chprof(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chprof), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chprof), z, sizeof(z));
  }
}
#endif

__device__ __forceinline__ void chains_window(ChainArgs const& A, int const a, u32* smem) {
  u32 const t = threadIdx.x, lane = t & 63u;
  GraphWs const& ws = A.ws;
  ma_params_t const& P = A.prm;
  int const w = static_cast<int>(ws.active[a]);
  u32 const n = ws.n_nodes[a];
  if ((ws.win_flags[w] & 4u) || n <= A.n_lo || n > A.cap || n == 0) return;  // not this launch's (k_clean reports overflows)
  u32 const cap = A.cap, xw = A.xw, kXeCap = A.xe_cap, kSegCap = A.seg_cap;
  int const S = ws.num_samples;
  u32 const NV = static_cast<u32>(S) + 2u, VS = 2u * xw;  // values per node, u16 stride of the value table
  u32 const K = static_cast<u32>(win_kmer(ws, w)), K1 = K - 1u;
  size_t const nb = static_cast<size_t>(a) * ws.nc;

  Lds L;
  {
    u32* p = smem;
    L.e2 = p; p += cap;
    L.x = p; p += cap * xw;
    L.blk = p; p += cap;
    L.key = p; p += cap;
    L.fl = reinterpret_cast<u16*>(p); p += cap / 2;
    L.a1 = reinterpret_cast<u16*>(p); p += cap / 2;
    L.segi = reinterpret_cast<u16*>(p); p += cap / 2;   // segi and p2n are adjacent: together the u32 component sizes
    L.p2n = reinterpret_cast<u16*>(p); p += cap / 2;
    L.abs = reinterpret_cast<u16*>(p); p += cap / 2;
    L.cid = reinterpret_cast<u16*>(p); p += cap / 2;
    L.xe = reinterpret_cast<u16*>(p); p += kXeCap * 4;
    L.seg_base = reinterpret_cast<u16*>(p); p += kSegCap / 2;
    L.seg_m = reinterpret_cast<u16*>(p); p += kSegCap / 2;
    L.seg_l = reinterpret_cast<u16*>(p); p += kSegCap / 2;
    L.seg_r = reinterpret_cast<u16*>(p); p += kSegCap / 2;
    L.seg_fl = reinterpret_cast<u16*>(p); p += kSegCap / 2;
    L.cl = reinterpret_cast<u16*>(p); p += A.cl_cap / 2;
    L.bmin = p; p += cap / 32;
    L.cand = p; p += 96;
    L.misc = p; p += 32;
  }
  u32* const csize = reinterpret_cast<u32*>(L.segi);
  if (t < 32) L.misc[t] = 0;
  __syncthreads();
#define PUNT() do { L.misc[M_PUNT] = static_cast<u32>(__LINE__); } while (0)
#define BAIL_IF_PUNT() do { __syncthreads(); if (L.misc[M_PUNT]) { if (threadIdx.x == 0) ws.cg_hdr[static_cast<size_t>(a) * kCgHdr + 3] = L.misc[M_PUNT]; return; } } while (0)

  CH_T0();
  // ---- the raw graph: flags, first two edges inline, the rare third and fourth in a side table ----
  for (u32 base = 0; base < n; base += 8 * kT) {  // eight nodes per thread: their loads in flight together
    u32 ne_[8], sg_[8], lb_[8];
    uint4 ev_[8];
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
      u32 const i = base + q * kT + t, ic = i < n ? i : 0u;
      ne_[q] = ws.nd_nedge[nb + ic];
      sg_[q] = ws.nd_sign[nb + ic];
      lb_[q] = ws.nd_label[nb + ic];
      ev_[q] = *reinterpret_cast<const uint4*>(ws.nd_edge + (nb + ic) * kEdgeCap);
    }
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
      u32 const i = base + q * kT + t;
      if (i >= n) continue;
      u32 const ne = ne_[q];
      uint4 const ev = ev_[q];
      u32 e1 = ne > 1 ? ev.y : 0u;
      if (ne > 8u) PUNT();
      if (ne > 2u) {
        u32 const idx = atomicAdd(&L.misc[M_XE], 1u);
        if (idx >= kXeCap) {
          PUNT();
        } else {
          L.xe[idx * 8 + 0] = static_cast<u16>(ev.y);
          L.xe[idx * 8 + 1] = static_cast<u16>(ev.z);
          L.xe[idx * 8 + 2] = static_cast<u16>(ev.w);
          if (ne > 4u && ne <= 8u) {  // rare (deep panels): the second quad
            uint4 const ev2 = *reinterpret_cast<const uint4*>(ws.nd_edge + (nb + i) * kEdgeCap + 4);
            L.xe[idx * 8 + 3] = static_cast<u16>(ev2.x);
            L.xe[idx * 8 + 4] = static_cast<u16>(ev2.y);
            L.xe[idx * 8 + 5] = static_cast<u16>(ev2.z);
            L.xe[idx * 8 + 6] = static_cast<u16>(ev2.w);
          }
          L.xe[idx * 8 + 7] = static_cast<u16>(ne);
        }
        e1 = idx;
      }
      L.e2[i] = (ne > 0 ? (ev.x & 0xFFFFu) : 0u) | (e1 << 16);
      L.fl[i] = static_cast<u16>((ne > 2u ? 3u : ne) | (sg_[q] ? F_SIGN : 0u) | ((lb_[q] & 7u) << F_LABEL_SH) | (15u << F_CAND_SH));
      L.x[i] = i;  // FastSV label
    }
  }
  BAIL_IF_PUNT();

  CH_ACC(0);
  // ---- MarkConnectedComponents (graph.cpp:392-463): FastSV hooking + shortcutting on LDS labels ----
  {
    u32* const lab = L.x;
    // Eight nodes per thread and trip: the loads of a level (label, parent's label, neighbours' labels, their parents')
    // are in flight together -- a node at a time every trip was five dependent LDS round trips.
    for (int round = 0; round < 64; ++round) {
      bool hooked = false;
      for (u32 base = 0; base < n; base += 8 * kT) {
        u32 ic[8], pu[8], gu[8], ev[8], ne[8], a0[8], a1[8];
        bool ok[8];
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          u32 const i = base + q * kT + t;
          ok[q] = i < n;
          ic[q] = ok[q] ? i : 0u;
          pu[q] = lab[ic[q]];
          ev[q] = L.e2[ic[q]];
          ne[q] = ok[q] ? (L.fl[ic[q]] & F_NE) : 0u;
        }
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          gu[q] = lab[pu[q]];
          a0[q] = lab[ne[q] >= 1u ? ((ev[q] & 0xFFFFu) >> 2) : ic[q]];
          a1[q] = lab[ne[q] == 2u ? (ev[q] >> 18) : ic[q]];
        }
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          a0[q] = lab[a0[q]];
          a1[q] = lab[a1[q]];
        }
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          u32 best = min(gu[q], min(a0[q], a1[q]));  // (a node without that edge contributed lab[lab[i]] = gu or larger)
          if (ne[q] > 2u) {
            u32 const nq = node_ne(L, ic[q]);
            for (u32 x = 1; x < nq; ++x) best = min(best, lab[lab[edge_at(L, ic[q], x) >> 2]]);
          }
          if (ok[q] && best < gu[q]) {
            atomicMin(&lab[pu[q]], best);
            atomicMin(&lab[ic[q]], best);
            hooked = true;
          }
        }
      }
      if (hooked) L.misc[M_CHANGED + (round & 1)] = 1u;
      __syncthreads();
      if (t == 0) L.misc[M_CHANGED + ((round + 1) & 1)] = 0;  // (read again only after the next round's second barrier)
      bool changed = false;
      for (u32 base = 0; base < n; base += 8 * kT) {
        u32 ic[8], pu[8], gu[8];
        bool ok[8];
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          u32 const i = base + q * kT + t;
          ok[q] = i < n;
          ic[q] = ok[q] ? i : 0u;
          pu[q] = lab[ic[q]];
        }
#pragma unroll
        for (u32 q = 0; q < 8; ++q) gu[q] = lab[pu[q]];
#pragma unroll
        for (u32 q = 0; q < 8; ++q)
          if (ok[q] && gu[q] < pu[q]) {
            lab[ic[q]] = gu[q];
            changed = true;
          }
      }
      if (changed) L.misc[M_CHANGED + (round & 1)] = 1u;
      __syncthreads();
      if (!L.misc[M_CHANGED + (round & 1)]) break;
    }
    CH_ACC(1);
    // component ids in discovery order = rank of the component's smallest node among the roots
    u32 const chunk = (n + kT - 1) / kT, i0 = t * chunk, i1 = min(n, i0 + chunk);
    u32 mine = 0;
    for (u32 i = i0; i < i1; ++i) mine += lab[i] == i ? 1u : 0u;
    u32 ncomp = 0;
    u32 rank = block_excl_scan(L, mine, &ncomp);
    for (u32 i = i0; i < i1; ++i)
      if (lab[i] == i) L.x[cap + i] = ++rank;
    if (t == 0) L.misc[M_NCOMP] = ncomp;
    __syncthreads();
    for (u32 i = t; i < n; i += kT) L.a1[i] = static_cast<u16>(L.x[cap + lab[i]]);
    __syncthreads();
  }
  u32 const ncomp = L.misc[M_NCOMP];

  CH_ACC(2);
  // ---- component sizes, FindSource / FindSink (graph.cpp:469-509) for every component at once ----
  const u32* refn = ws.ref_node + static_cast<size_t>(a) * ws.ref_stride;
  u32 const ref_len = A.b.ref_off[w + 1] - A.b.ref_off[w];
  u32 const n_refk = ref_len >= K + 1 ? ref_len - K + 1 : 0;
  {
    u32* const first_off = L.x;
    u32* const last_off = L.x + cap;
    for (u32 c = t; c <= ncomp; c += kT) {
      first_off[c] = 0xFFFFFFFFu;
      last_off[c] = 0;
      csize[c] = 0;
    }
    __syncthreads();
    for (u32 i = t; i < n; i += kT) atomicAdd(&csize[L.a1[i]], 1u);
    for (u32 base = 0; base < n_refk; base += 8 * kT) {
      u32 nd_[8], tot_[8];
#pragma unroll
      for (u32 q = 0; q < 8; ++q) {
        u32 const r = base + q * kT + t;
        nd_[q] = r < n_refk ? refn[r] : kNoNode;
      }
#pragma unroll
      for (u32 q = 0; q < 8; ++q) {
        tot_[q] = 0;
        if (nd_[q] != kNoNode)
          for (int s = 0; s < S; ++s) tot_[q] += ws.nd_cnt[(nb + nd_[q]) * S + s];
      }
#pragma unroll
      for (u32 q = 0; q < 8; ++q) {
        u32 const r = base + q * kT + t;
        if (nd_[q] == kNoNode || tot_[q] < P.min_anchor_cov) continue;
        u32 const c = L.a1[nd_[q]];
        atomicMin(&first_off[c], r);
        atomicMax(&last_off[c], r);
      }
    }
    __syncthreads();
    // candidates in component order, then stable by size descending (graph.cpp:441 made canonical)
    if (t < 64) {
      u32 ncand = 0;
      bool over = false;
      for (u32 c0 = 1; c0 <= ncomp; c0 += 64) {
        u32 const c = c0 + lane;
        bool ok = c <= ncomp && first_off[c] != 0xFFFFFFFFu;
        u32 so = 0, ko = 0, ns = 0, nk = 0;
        if (ok) {
          so = first_off[c];
          ko = last_off[c];
          ns = refn[so];
          nk = refn[ko];
          ok = ns != nk && ko - so + K >= static_cast<u32>(P.min_anchor_len);
        }
        unsigned long long const m = __ballot(ok);
        u32 const at = ncand + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)));
        if (ok) {
          if (at < static_cast<u32>(kCgMaxCand)) {
            L.cand[0 * 16 + at] = c;
            L.cand[1 * 16 + at] = csize[c];
            L.cand[2 * 16 + at] = ns;
            L.cand[3 * 16 + at] = nk;
            L.cand[4 * 16 + at] = so;
            L.cand[5 * 16 + at] = ko;
          } else {
            over = true;
          }
        }
        ncand += static_cast<u32>(__popcll(m));
      }
      if (__ballot(over)) PUNT();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0 && !L.misc[M_PUNT]) {
        for (u32 i = 1; i < ncand; ++i) {
          u32 j = i;
          while (j > 0 && L.cand[16 + j - 1] < L.cand[16 + j]) {
            for (u32 q = 0; q < 6; ++q) {
              u32 const tmp = L.cand[q * 16 + j];
              L.cand[q * 16 + j] = L.cand[q * 16 + j - 1];
              L.cand[q * 16 + j - 1] = tmp;
            }
            --j;
          }
        }
        L.misc[M_NCAND] = ncand;
      }
    }
  }
  BAIL_IF_PUNT();
  u32 const ncand = L.misc[M_NCAND];
  u32* const hdr = ws.cg_hdr + static_cast<size_t>(a) * kCgHdr;
  if (ncand == 0) {  // nothing to assemble at this k: the tail reports it
    if (t == 0) {
      hdr[0] = 0;
      hdr[1] = 0;
      ws.cg_state[a] = 1u;
    }
    return;
  }
  for (u32 i = t; i < n; i += kT) {
    u32 const c = L.a1[i];
    u32 ci = 15u, f = L.fl[i] & ~(15u << F_CAND_SH);
    for (u32 q = 0; q < ncand; ++q)
      if (L.cand[q] == c) ci = q;
    if (ci != 15u) {
      if (L.cand[2 * 16 + ci] == i) f |= F_SRC;
      if (L.cand[3 * 16 + ci] == i) f |= F_SNK;
    }
    L.fl[i] = static_cast<u16>(f | (ci << F_CAND_SH));
  }
  __syncthreads();

  CH_ACC(3);
  // ---- plain / fully plain ----
  for (u32 i = t; i < n; i += kT) {
    u32 const f = L.fl[i];
    if ((f >> F_CAND_SH) == 15u || (f & F_NE) != 2u) continue;
    u32 const v = L.e2[i], e0 = v & 0xFFFFu, e1 = v >> 16;
    bool const plain = (e0 >> 2) != i && (e1 >> 2) != i && (((e0 ^ e1) >> 1) & 1u);
    if (plain) L.fl[i] = static_cast<u16>(f | F_PLAIN);
  }
  __syncthreads();
  for (u32 i = t; i < n; i += kT) {
    u32 const f = L.fl[i];
    if (!(f & F_PLAIN) || (f & (F_SRC | F_SNK))) continue;
    u32 const v = L.e2[i], n0 = (v & 0xFFFFu) >> 2, n1 = v >> 18;
    if (n0 == n1) PUNT();  // a two-ring
    if ((L.fl[n0] & F_NE) <= 2u && (L.fl[n1] & F_NE) <= 2u) L.fl[i] = static_cast<u16>(f | F_INI);
  }
  BAIL_IF_PUNT();

  CH_ACC(4);
  // ---- chain states: (node, side it leaves by) -> next such state; list ranking by pointer jumping ----
  {
    u32* const st = L.x;  // state << 12 | hops
    for (u32 i = t; i < n; i += kT) {
      if (!(L.fl[i] & F_INI)) continue;
      for (u32 j = 0; j < 2; ++j) {
        u32 const e = side_edge(L, i, j), nx = e >> 2;
        st[2 * i + j] = (L.fl[nx] & F_INI) ? (((2 * nx + (e & 1u)) << 12) | 1u) : ((2 * i + j) << 12);
      }
    }
    // In-place, unsynchronised jumps are safe (a state's word is read and written whole, and only ever moves further along
    // its chain): one barrier a round, ceil(log2 n) + 1 rounds settle every path; a ring of plain nodes never does.
    int rounds = 2;
    while ((1u << (rounds - 1)) < n) ++rounds;
    for (int round = 0; round < rounds; ++round) {
      __syncthreads();
      for (u32 base = 0; base < n; base += 8 * kT) {
        u32 va[8], vb[8], wa[8], wb[8];
        bool ok[8];
        u32 ic[8];
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          u32 const i = base + q * kT + t;
          ok[q] = i < n && (L.fl[i < n ? i : 0u] & F_INI);
          ic[q] = ok[q] ? i : 0u;
        }
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          va[q] = ok[q] ? st[2 * ic[q]] : 0u;
          vb[q] = ok[q] ? st[2 * ic[q] + 1] : 0u;
        }
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          wa[q] = st[va[q] >> 12];
          wb[q] = st[vb[q] >> 12];
        }
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          if (!ok[q]) continue;
          if ((wa[q] >> 12) != (va[q] >> 12)) st[2 * ic[q]] = (wa[q] & ~0xFFFu) | ((va[q] & 0xFFFu) + (wa[q] & 0xFFFu));
          if ((wb[q] >> 12) != (vb[q] >> 12)) st[2 * ic[q] + 1] = (wb[q] & ~0xFFFu) | ((vb[q] & 0xFFFu) + (wb[q] & 0xFFFu));
        }
      }
    }
    __syncthreads();
    for (u32 i = t; i < n; i += kT) {
      if (!(L.fl[i] & F_INI)) continue;
      for (u32 j = 0; j < 2; ++j) {
        u32 const n1 = st[2 * i + j] >> 12;
        if ((st[n1] >> 12) != n1) PUNT();  // not at a terminal state yet: a ring
      }
    }
    BAIL_IF_PUNT();
    CH_ACC(5);
    // position 0 = the end with the smaller node index
    for (u32 i = t; i < n; i += kT) {
      if (!(L.fl[i] & F_INI)) continue;
      u32 const v0 = st[2 * i], v1 = st[2 * i + 1];
      u32 const e0 = v0 >> 13, e1 = v1 >> 13, d0 = v0 & 0xFFFu, d1 = v1 & 0xFFFu;
      if (e0 == e1 && d0 + d1 > 0) PUNT();
      bool const l0 = e0 <= e1;  // side 0 faces the L end
      u32 const pos = l0 ? d0 : d1;
      if (!l0) L.fl[i] = static_cast<u16>(L.fl[i] | F_SIDEL);
      L.a1[i] = static_cast<u16>(pos + 1u);
      L.segi[i] = static_cast<u16>(l0 ? e0 : e1);
      if (pos == 0) L.cid[i] = static_cast<u16>(d0 + d1 + 1u);
    }
    BAIL_IF_PUNT();
  }
  CH_ACC(6);
  // ---- segment table ----
  {
    u32 const chunk = (n + kT - 1) / kT, i0 = t * chunk, i1 = min(n, i0 + chunk);
    u32 mine = 0;
    for (u32 i = i0; i < i1; ++i) mine += ((L.fl[i] & F_INI) && L.a1[i] == 1u) ? 1u : 0u;
    u32 G = 0;
    u32 g = block_excl_scan(L, mine, &G);
    if (G > kSegCap) {
      if (t == 0) PUNT();
    } else {
      for (u32 i = i0; i < i1; ++i)
        if ((L.fl[i] & F_INI) && L.a1[i] == 1u) {
          L.seg_m[g] = L.cid[i];
          L.seg_l[g] = static_cast<u16>(i);  // (its first node, for now)
          L.key[i] = g;
          ++g;
        }
    }
    if (t == 0) L.misc[M_G] = G;
    BAIL_IF_PUNT();
    if (t < 64) {  // exclusive scan of the segment lengths
      u32 carry = 0;
      for (u32 g0 = 0; g0 < G; g0 += 64) {
        u32 const gg = g0 + lane;
        u32 const mm = gg < G ? L.seg_m[gg] : 0u;
        u32 inc = mm;
#pragma unroll
        for (u32 o = 1; o < 64; o <<= 1) {
          u32 const y = __shfl_up(inc, o, 64);
          if (lane >= o) inc += y;
        }
        if (gg < G) L.seg_base[gg] = static_cast<u16>(carry + inc - mm);
        carry += __shfl(inc, 63, 64);
      }
    }
    __syncthreads();
    for (u32 i = t; i < n; i += kT) {
      if (!(L.fl[i] & F_INI)) continue;
      u32 const gg = L.key[L.segi[i]];
      L.segi[i] = static_cast<u16>(gg);
      L.p2n[L.seg_base[gg] + L.a1[i] - 1u] = static_cast<u16>(i);
    }
    __syncthreads();
    // boundaries: L / R, do they pass the buddy test as the OPPOSITE neighbour (graph.cpp:758-799), are they tip ends,
    // which slot of their edge list points back into the segment
    for (u32 gg = t; gg < G; gg += kT) {
      u32 const base = L.seg_base[gg], m = L.seg_m[gg];
      u32 const c0 = L.p2n[base], cm = L.p2n[base + m - 1u];
      u32 const el = side_edge(L, c0, (L.fl[c0] & F_SIDEL) ? 1u : 0u), er = side_edge(L, cm, (L.fl[cm] & F_SIDEL) ? 0u : 1u);
      u32 const nl = el >> 2, nr = er >> 2;
      if (nl == nr) PUNT();
      u32 fl = 0;
      auto describe = [&](u32 B, u32 from, u32 e_in, u32 b_buddy, u32 b_tip, u32 slot_sh, u32 b_dm) {
        u32 const fb = L.fl[B], neb = fb & F_NE;
        if (e_in & 1u) fl |= b_dm;
        if (fb & F_PLAIN) {
          u32 const far = side_edge(L, B, e_in & 1u);  // B's back edge is on side !dst_minus: the far one on side dst_minus
          if ((L.fl[far >> 2] & F_NE) <= 2u) fl |= b_buddy;
        }
        if (neb == 1u && !(fb & (F_SRC | F_SNK))) fl |= b_tip;
        u32 const want = (from << 2) | kind_rev(e_in & 3u);
        u32 slot = 4;
        for (u32 x = 0; x < neb; ++x)
          if (edge_at(L, B, x) == want) slot = x;
        if (slot > 1u && neb <= 2u) PUNT();  // (the raw graph is mirror consistent: cannot happen)
        fl |= (slot & 3u) << slot_sh;
      };
      describe(nl, c0, el, SF_BL, SF_TIPL, SF_SLOTL_SH, SF_DML);
      describe(nr, cm, er, SF_BR, SF_TIPR, SF_SLOTR_SH, SF_DMR);
      L.seg_l[gg] = static_cast<u16>(nl);
      L.seg_r[gg] = static_cast<u16>(nr);
      L.seg_fl[gg] = static_cast<u16>(fl);
    }
    BAIL_IF_PUNT();
    // node indices ascending along the segment (k-mers enter the graph left to right: the reference backbone always is)?
    // then the minimum of any interval is its left end and the walker search below needs no range query
    for (u32 i = t; i < n; i += kT) {
      if (!(L.fl[i] & F_INI)) continue;
      u32 const gg = L.segi[i], pos = L.a1[i] - 1u;
      if (pos + 1u < L.seg_m[gg] && L.p2n[L.seg_base[gg] + pos + 1u] < i) atomicOr(reinterpret_cast<u32*>(L.seg_fl) + (gg >> 1), (gg & 1u) ? (SF_DESC << 16) : SF_DESC);
    }
    __syncthreads();
  }
  u32 const G = L.misc[M_G];

  CH_ACC(7);
  // ---- per-node values (per-sample counts, two role counts) as u16; blocks, absorbers, rewrite times ----
  u16* const val = reinterpret_cast<u16*>(L.x);
  for (u32 base = 0; base < n; base += 4 * kT) {
    u32 cv[4][16];
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      u32 const i = base + q * kT + t, ic = i < n ? i : 0u;
#pragma unroll
      for (u32 v = 0; v < 16; ++v) {
        cv[q][v] = 0;
        if (v < NV) cv[q][v] = v < static_cast<u32>(S) ? ws.nd_cnt[(nb + ic) * S + v] : ws.nd_role[(nb + ic) * 2 + (v - S)];
      }
    }
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      u32 const i = base + q * kT + t;
      if (i >= n) continue;
      bool const in = (L.fl[i] >> F_CAND_SH) != 15u;
#pragma unroll
      for (u32 v = 0; v < 16; ++v) {
        if (v >= VS) continue;
        u32 const c = in ? cv[q][v] : 0u;
        if (c > 0xFFFFu) PUNT();
        val[i * VS + v] = static_cast<u16>(c);
      }
      L.abs[i] = static_cast<u16>(kNone16);
      u32 const p1 = (L.fl[i] & F_INI) ? L.a1[i] : 0u;
      L.blk[i] = p1 | (p1 << 16);
      L.key[i] = 0;
    }
  }
  BAIL_IF_PUNT();

  CH_ACC(8);
  // ---- the interval process: one group of VL lanes per segment ----
  // Whose turn it is: CompressGraph visits nodes in index order, so the next walker of a segment is the remaining single
  // with the smallest node index -- among those whose turn can do anything: an interior single always absorbs one side;
  // position 0 (m - 1) only if L (R) is a buddy, otherwise its turn is a no-op whenever it comes and it can be left out
  // for good.  Every remaining candidate has a larger index than the last walker (or it would have walked first), so no
  // clock is needed; a tip end's turn comes when its index is smaller than that minimum, and ends the segment.
  // The minimum over the shrinking interval [a..b] is a range query: minima of the 32-entry blocks of p2n are prepared
  // once, a query reads the two partial blocks and the block minima in between (m / 128 + 16 trips instead of m / 4).
  {
    u32* const bmin = L.bmin;
    for (u32 bq = t; bq < (n + 31u) / 32u; bq += kT) {
      u32 best = 0xFFFFFFFFu;
      for (u32 q = 0; q < 32u; ++q) {
        u32 const idx = bq * 32u + q;
        if (idx < n) best = min(best, (static_cast<u32>(L.p2n[idx]) << 12) | idx);
      }
      bmin[bq] = best;  // (entries of p2n beyond the last segment are stale: such a block is never inside a query)
    }
    __syncthreads();
    u32 const VL = NV <= 4u ? 4u : (NV <= 8u ? 8u : 16u), GW = 64u / VL;
    u32 const v = t & (VL - 1u), q_ = t / VL, ngroups = kT / VL;
    u32 const gid = (q_ % GW) * 4u + q_ / GW;  // consecutive segments on different wavefronts
    bool const vlane = v < NV;
    auto grp_min = [&](u32 x) {
      for (u32 o = 1; o < VL; o <<= 1) x = min(x, static_cast<u32>(__shfl_xor(x, o, 64)));
      return x;
    };
    for (u32 gg = gid; gg < G; gg += ngroups) {
      u32 const base = L.seg_base[gg], m = L.seg_m[gg], nl = L.seg_l[gg], nr = L.seg_r[gg], sfl = L.seg_fl[gg];
      i32 a_ = 0, b_ = static_cast<i32>(m) - 1;   // remaining singles [a_..b_]
      i32 lbn = -1, rbn = -1;                      // owners of the blocks at the L / R end
      bool tip_l = (sfl & SF_TIPL) != 0, tip_r = (sfl & SF_TIPR) != 0;
      i32 const elig_lo = ((sfl & SF_BL) && m > 1u) ? 0 : 1, elig_hi = static_cast<i32>(m) - (((sfl & SF_BR) && m > 1u) ? 1 : 2);
      // running state of the walker whose turn it is
      u32 s = 0, len_a = 0, lab = 0, steps = 0;
      auto absorb = [&](u32 walker, u32 unit, bool single) {
        u32 sz = 1, len_b = K;
        if (!single) {
          u32 const bk = L.blk[unit];
          sz = (bk >> 16) - (bk & 0xFFFFu) + 1u;
          len_b = K1 + sz;
        }
        u32 const av = vlane ? val[unit * VS + v] : 0u;
        len_a += sz;  // node.cpp:91: the walker's length AFTER the merge
        s = div_floor(__umul24(s, len_a) + __umul24(av, len_b), len_a + len_b);
        lab |= L.fl[unit];
        if (v == 0) L.abs[unit] = static_cast<u16>(walker);
        ++steps;
      };
      // a run of `cnt` singles at positions p0, p0 + dir, ...: node ids, values and labels of eight of them are fetched
      // together (the loads do not depend on the running averages), so that the serial chain per step is multiply-add,
      // estimate, remainder, fix-up
      auto step1 = [&](u32 av) {
        len_a += 1u;
        u32 const den = len_a + K, num = __umul24(s, len_a) + __umul24(av, K);
        u32 qq = static_cast<u32>(static_cast<float>(num) * __builtin_amdgcn_rcpf(static_cast<float>(den)));
        i32 const rem = static_cast<i32>(num - __umul24(qq, den));
        if (rem < 0) --qq;
        else if (static_cast<u32>(rem) >= den) ++qq;
        s = qq;
      };
      auto absorb_run = [&](u32 walker, i32 p0, i32 cnt, i32 dir) {
        i32 c0 = 0;
        i32 const pb = static_cast<i32>(base) + p0;
        for (; c0 + 8 <= cnt; c0 += 8) {
          u32 ids[8], av[8];
#pragma unroll
          for (i32 q = 0; q < 8; ++q) ids[q] = L.p2n[pb + (c0 + q) * dir];
          float rr[8];
#pragma unroll
          for (i32 q = 0; q < 8; ++q) {
            av[q] = __umul24(vlane ? val[ids[q] * VS + v] : 0u, K);  // the absorbed k-mer's share of the numerator
            lab |= L.fl[ids[q]];  // (label bits picked out once the turn is over)
            rr[q] = __builtin_amdgcn_rcpf(static_cast<float>(len_a + static_cast<u32>(q) + 1u + K));
          }
          if (v == 0) {
#pragma unroll
            for (i32 q = 0; q < 8; ++q) L.abs[ids[q]] = static_cast<u16>(walker);
          }
          // the serial chain: multiply-add, estimate, remainder, fix-up -- nothing else depends on the running average
#pragma unroll
          for (i32 q = 0; q < 8; ++q) {
            len_a += 1u;
            u32 const den = len_a + K, num = __umul24(s, len_a) + av[q];
            u32 qq = static_cast<u32>(static_cast<float>(num) * rr[q]);
            i32 const rem = static_cast<i32>(num - __umul24(qq, den));
            qq += rem < 0 ? 0xFFFFFFFFu : (static_cast<u32>(rem) >= den ? 1u : 0u);
            s = qq;
          }
        }
        for (; c0 < cnt; ++c0) {
          u32 const id = L.p2n[pb + c0 * dir];
          u32 const av = vlane ? val[id * VS + v] : 0u;
          lab |= L.fl[id];
          if (v == 0) L.abs[id] = static_cast<u16>(walker);
          step1(av);
        }
        steps += static_cast<u32>(cnt);
      };
      auto begin_turn = [&](u32 x) {
        s = vlane ? val[x * VS + v] : 0u;
        len_a = K;
        lab = L.fl[x];
      };
      auto stamp = [&](u32 node, u32 slot, u32 walker, u32 pass) {
        if (v == 0) atomicMax(&L.key[node], (((walker << 13) | (pass << 12) | steps) << 1) | slot);
      };
      // min over p2n[base + lo .. base + hi] as id << 12 | (index - base)
      auto range_min = [&](i32 lo, i32 hi) {
        u32 best = 0xFFFFFFFFu;
        if (lo > hi) return best;
        u32 const g0 = base + static_cast<u32>(lo), g1 = base + static_cast<u32>(hi);
        u32 const h_end = min(g1, g0 | 31u);  // head: up to the end of g0's block
        for (u32 i = g0 + v; i <= h_end; i += VL) best = min(best, (static_cast<u32>(L.p2n[i]) << 12) | i);
        if (h_end < g1) {
          u32 const b0 = (h_end + 1u) >> 5, b1 = (g1 + 1u) >> 5;  // full blocks [b0, b1)
          for (u32 bq = b0 + v; bq < b1; bq += VL) best = min(best, bmin[bq]);
          for (u32 i = max(b1 << 5, h_end + 1u) + v; i <= g1; i += VL) best = min(best, (static_cast<u32>(L.p2n[i]) << 12) | i);
        }
        best = grp_min(best);
        return best == 0xFFFFFFFFu ? best : ((best & ~0xFFFu) | ((best & 0xFFFu) - base));
      };
      while (true) {
        CH_SUB0();
        i32 const qlo = max(a_, elig_lo), qhi = min(b_, elig_hi);
        u32 const best = (sfl & SF_DESC) ? range_min(qlo, qhi)
                                         : (qlo <= qhi ? ((static_cast<u32>(L.p2n[base + qlo]) << 12) | static_cast<u32>(qlo)) : 0xFFFFFFFFu);
        CH_SUB(12);
        u32 who = 0, bid = best >> 12;
        bool have = best != 0xFFFFFFFFu;
        if (tip_l && (!have || nl < bid)) {
          who = 1;
          bid = nl;
          have = true;
        }
        if (tip_r && (!have || nr < bid)) {
          who = 2;
          bid = nr;
          have = true;
        }
        if (!have) break;
        if (who != 0) {
          // a tip end swallows every unit of the segment, nearest first
          u32 const T = bid;
          u32 const side_t = ((L.e2[T] & 0xFFFFu) >> 1) & 1u;
          begin_turn(T);
          steps = 0;
          if (who == 1) {
            if (a_ > 0) absorb(T, static_cast<u32>(lbn), false);
            if (b_ >= a_) absorb_run(T, a_, b_ - a_ + 1, 1);
            if (b_ < static_cast<i32>(m) - 1) absorb(T, static_cast<u32>(rbn), false);
          } else {
            if (b_ < static_cast<i32>(m) - 1) absorb(T, static_cast<u32>(rbn), false);
            if (b_ >= a_) absorb_run(T, b_, b_ - a_ + 1, -1);
            if (a_ > 0) absorb(T, static_cast<u32>(lbn), false);
          }
          if (vlane) val[T * VS + v] = static_cast<u16>(s);
          if (v == 0) {
            u32 f = L.fl[T] & ~(F_SIDEL | (7u << F_LABEL_SH));
            f |= lab & (7u << F_LABEL_SH);
            // position -1 (L) / m (R); its one edge faces the segment: that is its R side / L side
            if (who == 1 ? (side_t == 0u) : (side_t == 1u)) f |= F_SIDEL;
            L.fl[T] = static_cast<u16>(f);
            L.a1[T] = static_cast<u16>(who == 1 ? 0u : m + 1u);
            L.segi[T] = static_cast<u16>(gg);
            L.blk[T] = who == 1 ? (0u | (m << 16)) : (1u | ((m + 1u) << 16));
          }
          stamp(T, 0u, T, 0u);
          stamp(who == 1 ? nr : nl, (sfl >> (who == 1 ? SF_SLOTR_SH : SF_SLOTL_SH)) & 3u, T, 0u);
          break;
        }
        i32 const j = static_cast<i32>(best & 0xFFFu);
        u32 const x = bid, fx = L.fl[x];
        u32 const side_r = (fx & F_SIDEL) ? 0u : 1u, side_l = side_r ^ 1u;
        bool const f_right = ((fx & F_SIGN) ? 0u : 1u) == side_r;  // CompressNode(x, dflt = true) first
        bool const has_l = j > 0, has_r = j < static_cast<i32>(m) - 1;
        bool walked_f = false, walked_any = false;
        begin_turn(x);
        i32 lo = j, hi = j;
        for (u32 pass = 0; pass < 2; ++pass) {
          bool const right = pass == 0 ? f_right : !f_right;
          if (!(right ? has_r : has_l)) continue;
          bool opp_ok;
          if (pass == 0) {
            opp_ok = (right ? has_l : has_r) || (sfl & (right ? SF_BL : SF_BR));
          } else {
            if ((right ? has_l : has_r) && !walked_f) continue;
            opp_ok = (sfl & (right ? SF_BL : SF_BR)) != 0;
          }
          if (!opp_ok) continue;
          steps = 0;
          if (right) {
            if (b_ > j) absorb_run(x, j + 1, b_ - j, 1);
            if (b_ < static_cast<i32>(m) - 1) absorb(x, static_cast<u32>(rbn), false);
            stamp(x, side_slot(L, x, side_r), x, pass);
            stamp(nr, (sfl >> SF_SLOTR_SH) & 3u, x, pass);
            b_ = j - 1;
            rbn = static_cast<i32>(x);
            hi = static_cast<i32>(m) - 1;
          } else {
            if (j > a_) absorb_run(x, j - 1, j - a_, -1);
            if (a_ > 0) absorb(x, static_cast<u32>(lbn), false);
            stamp(x, side_slot(L, x, side_l), x, pass);
            stamp(nl, (sfl >> SF_SLOTL_SH) & 3u, x, pass);
            a_ = j + 1;
            lbn = static_cast<i32>(x);
            lo = 0;
          }
          if (pass == 0) walked_f = true;
          walked_any = true;
        }
        CH_SUB(13);
        if (!walked_any) break;  // (cannot happen: every candidate's turn absorbs a side)
        if (lbn == static_cast<i32>(x) && rbn == static_cast<i32>(x)) {  // x holds the whole segment
          a_ = static_cast<i32>(m);
          b_ = static_cast<i32>(m) - 1;
          rbn = -1;
        }
        if (vlane) val[x * VS + v] = static_cast<u16>(s);
        if (v == 0) {
          L.fl[x] = static_cast<u16>((fx & ~(7u << F_LABEL_SH)) | (lab & (7u << F_LABEL_SH)));
          L.blk[x] = static_cast<u32>(lo + 1) | (static_cast<u32>(hi + 1) << 16);
        }
      }
    }
  }
  __syncthreads();

  CH_ACC(9);
  // ---- the compact graph ----
  u32 V = 0;
  {
    u32 const chunk = (n + kT - 1) / kT, i0 = t * chunk, i1 = min(n, i0 + chunk);
    u32 mine = 0;
    for (u32 i = i0; i < i1; ++i) mine += ((L.fl[i] >> F_CAND_SH) != 15u && L.abs[i] == kNone16) ? 1u : 0u;
    u32 c = block_excl_scan(L, mine, &V);
    if (V > ws.vc) {
      if (t == 0) PUNT();
    } else {
      for (u32 i = i0; i < i1; ++i)
        if ((L.fl[i] >> F_CAND_SH) != 15u && L.abs[i] == kNone16) {
          if (c < A.cl_cap) L.cl[c] = static_cast<u16>(i);
          L.cid[i] = static_cast<u16>(c++);
        }
    }
    BAIL_IF_PUNT();
  }
  size_t const vb = static_cast<size_t>(a) * ws.vc;
  u8* const pool = ws.cg_pool + static_cast<size_t>(a) * ws.pool_cap;
  bool const listed = V <= A.cl_cap;  // a thread per alive node when the list holds them all, else a sweep over the nodes
  for (u32 it = t; it < (listed ? V : n); it += kT) {
    u32 const i = listed ? static_cast<u32>(L.cl[it]) : it;
    u32 const f = L.fl[i], ci = f >> F_CAND_SH;
    if (ci == 15u || L.abs[i] != kNone16) continue;
    u32 const t = L.cid[i];  // (shadows the thread index: the record's slot)
    u32 const bk0 = L.blk[i];
    bool const my_owns = (bk0 >> 16) > (bk0 & 0xFFFFu);
    u32 const my_len = my_owns ? K1 + ((bk0 >> 16) - (bk0 & 0xFFFFu) + 1u) : K;
    for (int s = 0; s < S; ++s) ws.cg_cnt[(vb + t) * S + s] = val[i * VS + s];
    ws.cg_role[(vb + t) * 2] = val[i * VS + S];
    ws.cg_role[(vb + t) * 2 + 1] = val[i * VS + S + 1];
    ws.cg_label[vb + t] = static_cast<u8>((f >> F_LABEL_SH) & 7u);
    ws.cg_sign[vb + t] = (f & F_SIGN) ? 1 : 0;
    ws.cg_len[vb + t] = my_len;
    ws.cg_comp[vb + t] = ci + 1u;
    ws.cg_bsrc[vb + t] = my_owns ? 0x40000000u : ws.nd_src[nb + i];  // (an owner's pool offset is filled in below)
    ws.cg_blen[vb + t] = my_len;
    ws.cg_bsign[vb + t] = my_owns ? 1 : ((f & F_SIGN) ? 1 : 0);
    // edges: content from the final adjacency of the segments, order from the last rewrite
    u32 const ne = node_ne(L, i);
    u32 out_e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (u32 x = 0; x < 8; ++x) {
      if (x >= ne) continue;
      u32 const e = edge_at(L, i, x), d = e >> 2, sm = (e >> 1) & 1u;
      u32 ne_w = e;
      if (f & F_INI) {
        u32 const gg = L.segi[i], base = L.seg_base[gg], m = L.seg_m[gg], sfl = L.seg_fl[gg];
        u32 const bk = L.blk[i], lo = (bk & 0xFFFFu) - 1u, hi = (bk >> 16) - 1u;
        bool const left = sm == ((f & F_SIDEL) ? 1u : 0u);
        u32 nbn, ndm;
        if (left) {
          if (lo == 0) {
            nbn = L.seg_l[gg];
            ndm = (sfl & SF_DML) ? 1u : 0u;
          } else {
            nbn = top_owner(L, L.p2n[base + lo - 1u]);
            ndm = (L.fl[nbn] & F_SIDEL) ? 1u : 0u;           // it faces us with its R side: !sideR = sideL
          }
        } else {
          if (hi == m - 1u) {
            nbn = L.seg_r[gg];
            ndm = (sfl & SF_DMR) ? 1u : 0u;
          } else {
            nbn = top_owner(L, L.p2n[base + hi + 1u]);
            ndm = (L.fl[nbn] & F_SIDEL) ? 0u : 1u;           // it faces us with its L side
          }
        }
        ne_w = (nbn << 2) | (sm << 1) | ndm;
      } else if (L.fl[d] & F_INI) {  // d is the extremity of a segment this boundary node closes
        u32 const gg = L.segi[d], sfl = L.seg_fl[gg];
        bool const is_l = L.seg_l[gg] == i;
        if (my_owns) {  // a tip end that swallowed the segment: its edge now reaches the far boundary
          u32 const far = is_l ? L.seg_r[gg] : L.seg_l[gg];
          ne_w = (far << 2) | (sm << 1) | ((sfl & (is_l ? SF_DMR : SF_DML)) ? 1u : 0u);
        } else {
          u32 const u = top_owner(L, d);
          u32 const sl = (L.fl[u] & F_SIDEL) ? 1u : 0u;
          ne_w = (u << 2) | (sm << 1) | (is_l ? (sl ^ 1u) : sl);
        }
      }
      out_e[x] = ((static_cast<u32>(L.cid[ne_w >> 2])) << 2) | (ne_w & 3u);
    }
    u32 const kk = L.key[i];
    if (ne == 2u && kk != 0u && (kk & 1u) == 0u) {  // slot 0 was rewritten last: it moved behind slot 1
      u32 const tmp = out_e[0];
      out_e[0] = out_e[1];
      out_e[1] = tmp;
    }
    ws.cg_nedge[vb + t] = static_cast<u8>(ne);
#pragma unroll
    for (u32 x = 0; x < 8; ++x)
      if (x < 4u || ne > 4u) ws.cg_edge[(vb + t) * kCgEdgeCap + x] = out_e[x];
  }
  CH_ACC(10);
  // merged strings: one pool region per top-level owner, in node order; the offset lives where the rewrite time did
  {
    __syncthreads();
    u32 const chunk = (n + kT - 1) / kT, i0 = t * chunk, i1 = min(n, i0 + chunk);
    auto owner_len = [&](u32 i) -> u32 {
      if ((L.fl[i] >> F_CAND_SH) == 15u || L.abs[i] != kNone16) return 0u;
      u32 const bk = L.blk[i];
      return (bk >> 16) > (bk & 0xFFFFu) ? K1 + ((bk >> 16) - (bk & 0xFFFFu) + 1u) : 0u;
    };
    u32 mine = 0;
    for (u32 i = i0; i < i1; ++i) mine += owner_len(i);
    u32 total = 0;
    u32 off = block_excl_scan(L, mine, &total);
    if (total > ws.pool_cap || total > 0x3FFFFFFFu) {
      if (t == 0) PUNT();
    } else {
      for (u32 i = i0; i < i1; ++i) {
        u32 const ln = owner_len(i);
        if (ln == 0) continue;
        L.key[i] = off;
        ws.cg_bsrc[vb + L.cid[i]] = 0x40000000u | off;
        off += ln;
      }
    }
    if (t == 0) L.misc[M_POOL] = total;
    BAIL_IF_PUNT();
  }
  // leaf bytes -> pool
  {
    const u8* refb = A.b.ref_bases + A.b.ref_off[w];
    const u8* readb = A.b.read_bases + A.b.read_off[A.b.read_win_off[w]];
    auto pos1 = [&](u32 q) { return static_cast<u32>(L.a1[q]); };
    auto sideL = [&](u32 q) { return (L.fl[q] & F_SIDEL) ? 1u : 0u; };
    auto prepend_count = [&](u32 y) {
      u32 const bk = L.blk[y];
      return sideL(y) == 0u ? (bk >> 16) - pos1(y) : pos1(y) - (bk & 0xFFFFu);  // sideR == 1 <=> sideL == 0
    };
    auto place = [&](u32 y, i32 idx, u8 base, bool comp) {
      while (L.abs[y] != kNone16) {
        u32 const z = L.abs[y];
        u32 const bk = L.blk[y], lo1 = bk & 0xFFFFu, hi1 = bk >> 16;
        i32 const sz = static_cast<i32>(hi1 - lo1 + 1u), LY = static_cast<i32>(K1) + sz;
        bool const right = lo1 > pos1(z);
        u32 const szb = right ? (sideL(z) ^ 1u) : sideL(z);
        u32 const j = (right ? sideL(y) : (sideL(y) ^ 1u)) ^ 1u;
        bool const rc = szb != j, append = szb == 0u;
        i32 const d1 = right ? static_cast<i32>(lo1 - pos1(z)) : static_cast<i32>(pos1(z) - hi1), d2 = d1 + sz - 1;
        i32 const PZ = static_cast<i32>(prepend_count(z));
        i32 const vv = rc ? LY - 1 - idx : idx;
        if (append) {
          if (vv < static_cast<i32>(K1)) return;
          idx = PZ + static_cast<i32>(K) + d1 - 1 + (vv - static_cast<i32>(K1));
        } else {
          if (vv > sz - 1) return;
          idx = PZ - d2 + vv;
        }
        comp ^= rc;
        y = z;
      }
      pool[L.key[y] + static_cast<u32>(idx)] = comp ? dev_complement(base) : base;
    };
    auto src_of = [&](u32 q) -> const u8* {
      u32 const sv = ws.nd_src[nb + q];
      return (sv & 0x80000000u) ? readb + (sv & 0x7FFFFFFFu) : refb + sv;
    };
    // an absorbed k-mer leaves one base
    for (u32 q = t; q < n; q += kT) {
      u32 const f = L.fl[q];
      if ((f >> F_CAND_SH) == 15u || L.abs[q] == kNone16) continue;
      u32 const bk = L.blk[q];
      if ((bk >> 16) > (bk & 0xFFFFu)) continue;  // a block: its bytes travel with its owner's
      const u8* p = src_of(q);
      bool const plus = (f & F_SIGN) != 0;
      u32 const y = L.abs[q];
      bool const right = pos1(q) > pos1(y);
      u32 const sy = right ? (sideL(y) ^ 1u) : sideL(y);
      u32 const j = (right ? sideL(q) : (sideL(q) ^ 1u)) ^ 1u;
      bool const rc = sy != j, append = sy == 0u;
      i32 const d = right ? static_cast<i32>(pos1(q) - pos1(y)) : static_cast<i32>(pos1(y) - pos1(q));
      i32 const PP = static_cast<i32>(prepend_count(y));
      u32 const xx = (append != rc) ? K1 : 0u;  // append: last base of the oriented k-mer, prepend: first; rc flips which canonical base that is
      u8 const base = plus ? p[xx] : dev_complement(p[K1 - xx]);
      place(y, append ? PP + static_cast<i32>(K) + d - 1 : PP - d, base, rc);
    }
    // a top-level owner's k-mer: (owner, byte) pairs over all threads when the alive list holds the graph
    if (listed) {
      for (u32 it = t; it < V * K; it += kT) {
        u32 const q = L.cl[it / K], i = it % K;
        u32 const bk = L.blk[q];
        if ((bk >> 16) <= (bk & 0xFFFFu)) continue;
        const u8* p = src_of(q);
        u8 const base = (L.fl[q] & F_SIGN) ? p[i] : dev_complement(p[K1 - i]);
        place(q, static_cast<i32>(prepend_count(q)) + static_cast<i32>(i), base, false);
      }
    }
    // nested owners (and every owner when there is no list): a thread per owner
    for (u32 q = t; q < n; q += kT) {
      u32 const f = L.fl[q];
      if ((f >> F_CAND_SH) == 15u || (listed && L.abs[q] == kNone16)) continue;
      u32 const bk = L.blk[q];
      if ((bk >> 16) <= (bk & 0xFFFFu)) continue;
      const u8* p = src_of(q);
      bool const plus = (f & F_SIGN) != 0;
      i32 const PP = static_cast<i32>(prepend_count(q));
      for (u32 i = 0; i < K; ++i) place(q, PP + static_cast<i32>(i), plus ? p[i] : dev_complement(p[K1 - i]), false);
    }
  }
  CH_ACC(11);
  if (t == 0) {
    hdr[0] = V;
    hdr[1] = ncand;
    hdr[2] = L.misc[M_POOL];  // bytes of merged strings in the pool
    for (u32 q = 0; q < ncand; ++q) {
      u32* c = hdr + 8 + 6 * q;
      c[0] = q + 1u;
      c[1] = L.cand[1 * 16 + q];
      c[2] = L.cid[L.cand[2 * 16 + q]];
      c[3] = L.cid[L.cand[3 * 16 + q]];
      c[4] = L.cand[4 * 16 + q];
      c[5] = L.cand[5 * 16 + q];
    }
    __threadfence();
    ws.cg_state[a] = 1u;
  }
#undef PUNT
#undef BAIL_IF_PUNT
}

// One workgroup per window for the common size class; the two larger classes (2.5 kb windows, deep panels: 81 / 144 KB of
// LDS) are launched as a few hundred workgroups that walk the batch -- two thousand workgroups that each claim most of a CU's
// LDS only to find that their window belongs to another launch held the lane up behind the other lanes' kernels.
__global__ __launch_bounds__(kT) void k_clean_chains(ChainArgs A) {
  extern __shared__ u32 smem[];
  chains_window(A, static_cast<int>(blockIdx.x), smem);
}
__global__ __launch_bounds__(kT) void k_clean_chains_walk(ChainArgs A) {
  extern __shared__ u32 smem[];
  for (int a = blockIdx.x; a < A.ws.n_active; a += gridDim.x) {
    chains_window(A, a, smem);
    __syncthreads();
  }
}

int run_clean_chains(ma_ctx* ctx, const DBatch& b, const GraphWs& ws, const ma_params_t& prm) {
  u32 const S = static_cast<u32>(ws.num_samples);
  u32 const xw = std::max<u32>(2u, (S + 2u + 1u) / 2u);
  auto lds_bytes = [&](u32 cap) {
    u32 const kXeCap = cap <= 1472u ? 64u : cap / 8u, kSegCap = cap <= 1472u ? 256u : cap / 4u;
    size_t words = static_cast<size_t>(cap) * (3 + xw) + static_cast<size_t>(cap) * 3 /* six u16 arrays */ + kXeCap * 4 + (kSegCap / 2) * 5 + 96 + 32 + cap / 32 + (cap <= 1472u ? 256u : cap / 4u) / 2;
    return words * 4;
  };
  // (per call, like every other launch of more than 64 KB: the attribute belongs to the device the context runs on)
  MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_clean_chains), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_clean_chains_walk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // three launches by graph size: LDS is handed out in coarse granules, so the image of the common case is cut to fit
  // THREE workgroups per CU with room to spare (1472 nodes: 51.9 KB; at 1536 nodes only two fitted and the kernel took
  // 2.18 instead of 1.58 ms), the next class two per CU, the deepest windows one.  A workgroup whose window belongs to
  // another launch returns at once (0.02 ms per launch).
  u32 caps[3] = {1472u, 2304u, 4096u};
  if (const char* e = getenv("MA_CHAINS_CAP")) caps[0] = static_cast<u32>(atoi(e)) / 64u * 64u;
  u32 lo = 0;
  for (int l = 0; l < 3; ++l) {
    if (caps[l] <= lo || lds_bytes(caps[l]) > 160 * 1024) continue;
    ChainArgs args{b, ws, prm, caps[l], lo, xw, caps[l] <= 1472u ? 64u : caps[l] / 8u, caps[l] <= 1472u ? 256u : caps[l] / 4u, caps[l] <= 1472u ? 256u : caps[l] / 4u,
                   getenv("MA_CHAINS_STOP") ? static_cast<u32>(atoi(getenv("MA_CHAINS_STOP"))) : 0u};
    ctx->tic("k_clean_chains");
    if (l == 0)
      hipLaunchKernelGGL(k_clean_chains, dim3(ws.n_active), dim3(kT), lds_bytes(caps[l]), ctx->stream, args);
    else
      hipLaunchKernelGGL(k_clean_chains_walk, dim3(std::min<u32>(static_cast<u32>(ws.n_active), 256u)), dim3(kT), lds_bytes(caps[l]), ctx->stream, args);
    ctx->toc();
    lo = caps[l];
  }
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

}  // namespace ma
