// Haplotype <-> reference partial-order alignment + variant extraction on gfx950.
//
// Replaces caller::MsaBuilder::UpdateSpoaState (caller/msa_builder.cpp:29-42; SPOA 4.1.5 engine
// kNW with convex gaps 0/-6/-6,-2/-26,-1: caller/msa_builder.h:72-89) and the caller::VariantSet
// constructor (caller/variant_extractor.cpp:24-233, variant_bubble.cpp:16-116, raw_variant.cpp:44-77).
//
// One wavefront per window; the window's components are processed one after another.
//  * The POA graph (node chars, in/out adjacency with 16-bit ids, per-edge haplotype label masks,
//    aligned-node rings, rank <-> node maps, DFS scratch, the alignment path) lives in LDS: the graph
//    update (Graph::AddAlignment), SPOA's DFS topological sort and the bubble walk are pointer-chasing
//    serial loops, so they run at LDS latency on lane 0 instead of HBM latency.
//  * The sequence-to-DAG DP keeps SPOA's five i32 matrices (H,F,E,O,Q) in HBM because the traceback
//    compares VALUES.  All 64 lanes fill them as a skewed pipeline: lane l owns a contiguous chunk of
//    columns and handles DP row (t - l + 1) at step t; the (H,E,Q) of the column to its left arrive
//    from lane l-1 by wave shuffle; the previous row stays in registers, so a row whose only
//    predecessor is the previous rank (the common case) reads nothing from HBM.
// Edge weights are not tracked: the reference only reads topology, labels and ranks from the POA
// graph (variant_extractor.cpp:47-58, :84-94, :159-181).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ma_internal.h"

namespace ma {

namespace {

constexpr int kPE = 4;     // in / out edges and aligned nodes kept per POA node
constexpr i32 kNegInf = static_cast<i32>(0x80000000u) + 1024;
constexpr i32 M_ = 0, N_ = -6, G_ = -6, E_ = -2, Q_ = -26, C_ = -1;  // msa_builder.h:72-77
constexpr u16 kNone16 = 0xFFFFu;

struct PoaWs {
  u32 pn;         // node capacity (LDS)
  u32 max_l;
  size_t cells;   // DP cells per matrix per window (skewed body)
  u32 cw_max;     // columns per lane for the longest haplotype (multiple of 4)
  i16* H;         // [w][cells] saturated i16
  i16* F;
  i16* E;
  i16* O;
  i16* Q;
  i32* C0;        // [w][5][pn + 2] column 0 of H, F, E, O, Q
};

// All graph arrays live in the kernel's dynamic LDS.  They are addressed as offsets into the
// __shared__ array (NOT through generic pointers kept in a struct: those compile to flat_load + pointer
// reloads from scratch instead of ds_read).
extern __shared__ unsigned char ma_lds[];
template <class T>
struct LdsArr {
  u32 off;  // byte offset into ma_lds
  __device__ __forceinline__ T& operator[](u32 i) const { return *reinterpret_cast<T*>(&ma_lds[off + i * sizeof(T)]); }
};

struct PG {  // LDS-resident POA graph of one window
  u32 nn, nseq, nrank;
  u32 pn;
  LdsArr<u8> nchar, nin, nout, nal, marks, ignored;
  LdsArr<u16> in_tail, out_head, out_lab, al, rank2node, node2rank;
  LdsArr<u16> stack;  // DFS stack; aliased by the column-0 DP values during alignment
  u32 stack_cap;
  i32 seq_first[16];
  bool overflow;
};

__device__ i32 pg_add_node(PG& g, u8 ch) {
  if (g.nn >= g.pn) {
    g.overflow = true;
    return 0;
  }
  u32 const id = g.nn++;
  g.nchar[id] = ch;
  g.nin[id] = g.nout[id] = g.nal[id] = 0;
  return static_cast<i32>(id);
}
__device__ void pg_add_edge(PG& g, u32 tail, u32 head) {  // spoa::Graph::AddEdge (weights dropped)
  u16 const lab = static_cast<u16>(1u << g.nseq);
  for (int x = 0; x < g.nout[tail]; ++x)
    if (g.out_head[tail * kPE + x] == head) {
      g.out_lab[tail * kPE + x] |= lab;
      return;
    }
  if (g.nout[tail] >= kPE || g.nin[head] >= kPE) {
    g.overflow = true;
    return;
  }
  g.out_head[tail * kPE + g.nout[tail]] = static_cast<u16>(head);
  g.out_lab[tail * kPE + g.nout[tail]] = lab;
  g.nout[tail]++;
  g.in_tail[head * kPE + g.nin[head]++] = static_cast<u16>(tail);
}
__device__ i32 pg_add_sequence(PG& g, const u8* seq, u32 begin, u32 end) {  // spoa::Graph::AddSequence
  if (begin == end) return -1;
  i32 prev = -1;
  u32 const first = g.nn;
  for (u32 i = begin; i < end; ++i) {
    i32 const cur = pg_add_node(g, seq[i]);
    if (g.overflow) return -1;
    if (prev >= 0) pg_add_edge(g, static_cast<u32>(prev), static_cast<u32>(cur));
    prev = cur;
  }
  return static_cast<i32>(first);
}
__device__ i32 pg_successor(const PG& g, u32 node, u32 label) {  // spoa::Graph::Node::Successor
  for (int x = 0; x < g.nout[node]; ++x)
    if (g.out_lab[node * kPE + x] & (1u << label)) return static_cast<i32>(g.out_head[node * kPE + x]);
  return -1;
}
// spoa::Graph::TopologicalSort
__device__ void pg_toposort(PG& g) {
  for (u32 i = 0; i < g.nn; ++i) g.marks[i] = g.ignored[i] = 0;
  g.nrank = 0;
  LdsArr<u16> const stack = g.stack;
  for (u32 s = 0; s < g.nn; ++s) {
    if (g.marks[s] != 0) continue;
    u32 sp = 0;
    stack[sp++] = static_cast<u16>(s);
    while (sp > 0) {
      u32 const cur = stack[sp - 1];
      bool valid = true;
      if (g.marks[cur] != 2) {
        for (int x = 0; x < g.nin[cur]; ++x) {
          u32 const t = g.in_tail[cur * kPE + x];
          if (g.marks[t] != 2) {
            if (sp < g.stack_cap) stack[sp++] = static_cast<u16>(t); else g.overflow = true;
            valid = false;
          }
        }
        if (!g.ignored[cur]) {
          for (int x = 0; x < g.nal[cur]; ++x) {
            u32 const an = g.al[cur * kPE + x];
            if (g.marks[an] != 2) {
              if (sp < g.stack_cap) stack[sp++] = static_cast<u16>(an); else g.overflow = true;
              g.ignored[an] = 1;
              valid = false;
            }
          }
        }
        if (valid) {
          g.marks[cur] = 2;
          if (!g.ignored[cur]) {
            g.rank2node[g.nrank++] = static_cast<u16>(cur);
            for (int x = 0; x < g.nal[cur]; ++x) g.rank2node[g.nrank++] = g.al[cur * kPE + x];
          }
        } else {
          g.marks[cur] = 1;
        }
      }
      if (valid) sp--;
      if (g.overflow) return;
    }
  }
  for (u32 r = 0; r < g.nrank; ++r) g.node2rank[g.rank2node[r]] = static_cast<u16>(r);
}

// spoa::Graph::AddAlignment; aln pairs are (node id + 1 | 0, seq pos + 1 | 0) in LDS
__device__ void pg_add_alignment(PG& g, LdsArr<u16> aln, u32 naln, const u8* seq, u32 len) {
  if (len == 0) return;
  if (naln == 0) {
    i32 const first = pg_add_sequence(g, seq, 0, len);
    g.seq_first[g.nseq++] = first;
    if (!g.overflow) pg_toposort(g);
    return;
  }
  i32 vfront = -1, vback = -1;
  for (u32 x = 0; x < naln; ++x)
    if (aln[2 * x + 1] != 0) {
      if (vfront < 0) vfront = static_cast<i32>(aln[2 * x + 1]) - 1;
      vback = static_cast<i32>(aln[2 * x + 1]) - 1;
    }
  i32 begin = pg_add_sequence(g, seq, 0, static_cast<u32>(vfront));
  i32 prev = begin >= 0 ? static_cast<i32>(g.nn - 1) : -1;
  i32 const last = pg_add_sequence(g, seq, static_cast<u32>(vback) + 1, len);
  for (u32 x = 0; x < naln && !g.overflow; ++x) {
    if (aln[2 * x + 1] == 0) continue;
    u32 const sp = static_cast<u32>(aln[2 * x + 1]) - 1;
    u8 const ch = seq[sp];
    i32 curr = -1;
    if (aln[2 * x] == 0) {
      curr = pg_add_node(g, ch);
    } else {
      u32 const jt = static_cast<u32>(aln[2 * x]) - 1;
      if (g.nchar[jt] == ch) {
        curr = static_cast<i32>(jt);
      } else {
        for (int y = 0; y < g.nal[jt]; ++y)
          if (g.nchar[g.al[jt * kPE + y]] == ch) {
            curr = static_cast<i32>(g.al[jt * kPE + y]);
            break;
          }
        if (curr < 0) {
          curr = pg_add_node(g, ch);
          if (g.overflow) break;
          int const na = g.nal[jt];
          if (na + 1 > kPE) {
            g.overflow = true;
            break;
          }
          for (int y = 0; y < na; ++y) {
            u32 const kt = g.al[jt * kPE + y];
            g.al[kt * kPE + g.nal[kt]++] = static_cast<u16>(curr);
            g.al[static_cast<u32>(curr) * kPE + g.nal[curr]++] = static_cast<u16>(kt);
          }
          g.al[jt * kPE + g.nal[jt]++] = static_cast<u16>(curr);
          g.al[static_cast<u32>(curr) * kPE + g.nal[curr]++] = static_cast<u16>(jt);
        }
      }
    }
    if (g.overflow) break;
    if (begin < 0) begin = curr;
    if (prev >= 0) pg_add_edge(g, static_cast<u32>(prev), static_cast<u32>(curr));
    prev = curr;
  }
  if (last >= 0 && prev >= 0) pg_add_edge(g, static_cast<u32>(prev), static_cast<u32>(last));
  g.seq_first[g.nseq++] = begin;
  if (!g.overflow) pg_toposort(g);
}

__device__ i32 classify_variant(const u8* r, u32 rl, const u8* a, u32 al) {  // raw_variant.cpp:44-77
  u32 s = 0;
  while (s < rl && s < al && r[s] == a[s]) s++;
  if (s == rl && s == al) return -1;
  u32 e = 0;
  while (e < (rl - s) && e < (al - s) && r[rl - 1 - e] == a[al - 1 - e]) e++;
  u32 const rc = rl - s - e, ac = al - s - e;
  if (rc == 0 && ac > 0) return 1;
  if (rc > 0 && ac == 0) return 2;
  if (rc == 0 || ac == 0) return -1;
  if (rc != ac) return 4;
  return rc == 1 ? 0 : 3;
}
__device__ i32 variant_length(const u8* r, u32 rl, const u8* a, u32 al, i32 t) {  // variant_bubble.cpp:16-47
  if (t == 0) return 1;
  i32 const R = static_cast<i32>(rl), A = static_cast<i32>(al);
  if (t == 1 || t == 2 || t == 4) return A - R;
  i32 s = 0;
  while (s < R && s < A && r[s] == a[s]) s++;
  i32 e = 0;
  while (e < (R - s) && e < (A - s) && r[R - 1 - e] == a[A - 1 - e]) e++;
  return A - s - e;
}
__device__ int bytes_cmp(const u8* a, u32 al, const u8* b, u32 bl) {  // std::string operator<=>
  u32 const m = al < bl ? al : bl;
  for (u32 i = 0; i < m; ++i)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return al < bl ? -1 : (al > bl ? 1 : 0);
}


// predecessor DP row x of DP row i (row = rank + 1), in in-edge order
__device__ __forceinline__ u32 row_pred(const PG& g, u32 node, u32 x) {
  return static_cast<u32>(g.node2rank[g.in_tail[node * kPE + x]]) + 1u;
}

// DP matrices are stored SKEWED so that the pipeline's stores coalesce: cell (i, j >= 1) belongs to lane
// l = (j-1)/cw, c = (j-1)%cw and lives at (i + l) * 64*cw + l*cw + c -- at pipeline step t every lane
// works on i + l == t + 1, so one store instruction of the wave covers 64 * 16 B of contiguous HBM.
// Column 0 of the five matrices is kept in five small side arrays.
// Bodies are stored as saturated i16: every value on or next to an optimal path is a real score
// (>= -6 * max_hap_len - 26 > -32768), and "minus infinity" cells only ever lose max() comparisons and
// equality tests, so clamping them to -32768 cannot change the traceback (column 0 stays i32).
struct DP {
  i16 *H, *F, *E, *O, *Q;      // skewed bodies
  i32 *H0, *F0, *E0, *O0, *Q0; // column 0, [V + 1]
  u32 cw, rs;                  // columns per lane (multiple of 4), row stride = 64 * cw
  __device__ __forceinline__ size_t off(u32 i, u32 j) const {  // j >= 1
    u32 const l = (j - 1) / cw;
    return static_cast<size_t>(i + l) * rs + (j - 1);
  }
  // column 0 holds true i32 values; clamp on read so both sources compare on the same scale
  __device__ __forceinline__ static i32 c0(i32 v) { return v < -32768 ? -32768 : v; }
  __device__ __forceinline__ i32 h(u32 i, u32 j) const { return j ? H[off(i, j)] : c0(H0[i]); }
  __device__ __forceinline__ i32 f(u32 i, u32 j) const { return j ? F[off(i, j)] : c0(F0[i]); }
  __device__ __forceinline__ i32 e(u32 i, u32 j) const { return j ? E[off(i, j)] : c0(E0[i]); }
  __device__ __forceinline__ i32 o(u32 i, u32 j) const { return j ? O[off(i, j)] : c0(O0[i]); }
  __device__ __forceinline__ i32 q(u32 i, u32 j) const { return j ? Q[off(i, j)] : c0(Q0[i]); }
};

// Skewed-pipeline fill of the five DP matrices (SisdAlignmentEngine::Convex, alignment phase).
// Fast path (L <= 64 * CWM): the row a lane finished in the previous step stays in registers.
constexpr int CWM = 16;
__device__ __forceinline__ u32 sat_pack(i32 a, i32 b) {  // two saturated i16 in one register
  a = a < -32768 ? -32768 : a;
  b = b < -32768 ? -32768 : b;
  return (static_cast<u32>(a) & 0xFFFFu) | (static_cast<u32>(b) << 16);
}
__device__ void poa_fill(const PG& g, const DP& d, u32 V, u32 L, int lane, const u8* seq, LdsArr<i32> col0) {
  u32 const cw = d.cw;
  u32 const jb = 1 + lane * cw;
  u32 const je = min(L + 1, jb + cw);
  u32 const nl = (L + cw - 1) / cw;
  u32 scp[CWM / 4];  // the lane's sequence chars, packed 4 per register
#pragma unroll
  for (int c4 = 0; c4 < CWM / 4; ++c4) {
    u32 pk = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) pk |= static_cast<u32>((jb + c4 * 4 + b < je) ? seq[jb + c4 * 4 + b - 1] : 0) << (8 * b);
    scp[c4] = pk;
  }
  i32 pH[CWM], pF[CWM], pO[CWM];
  i32 pHl = 0;
#pragma unroll
  for (int c = 0; c < CWM; ++c) pH[c] = pF[c] = pO[c] = 0;
  i32 hl = 0, el = 0, ql = 0;
  for (u32 t = 0; t + 1 < V + nl; ++t) {
    i32 const hL = __shfl_up(hl, 1), eL = __shfl_up(el, 1), qL = __shfl_up(ql, 1);
    i32 const row = static_cast<i32>(t) - lane + 1;
    bool const work = lane < static_cast<int>(nl) && row >= 1 && row <= static_cast<i32>(V);
    u32 node = 0, np = 0;
    bool needs_hbm = false;
    if (work) {
      node = g.rank2node[row - 1];
      np = g.nin[node];
      if (np == 0) needs_hbm = true;
      for (u32 x = 0; x < np; ++x) needs_hbm |= !(row_pred(g, node, x) + 1 == static_cast<u32>(row) && row >= 2);
    }
    // rows read back from HBM were written by other lanes of this wave >= 2 steps ago: make them visible
    if (__any(needs_hbm)) __threadfence_block();
    if (work) {
      u32 const i = static_cast<u32>(row);
      u32 const nch = g.nchar[node];
      i32 f[CWM], o[CWM], hm[CWM];
      for (u32 x = 0; x < (np ? np : 1u); ++x) {
        u32 const pr = np ? row_pred(g, node, x) : 0u;
        if (pr + 1 == i && i >= 2) {  // previous rank: still in registers
          i32 hprev = pHl;
#pragma unroll
          for (int c = 0; c < CWM; ++c) {
            i32 const mc = (nch == ((scp[c >> 2] >> (8 * (c & 3))) & 0xFFu)) ? M_ : N_;
            i32 const fv = max(pH[c] + G_, pF[c] + E_);
            i32 const ov = max(pH[c] + Q_, pO[c] + C_);
            i32 const hv = hprev + mc;
            hprev = pH[c];
            if (x == 0) {
              f[c] = fv;
              o[c] = ov;
              hm[c] = hv;
            } else {
              f[c] = max(f[c], fv);
              o[c] = max(o[c], ov);
              hm[c] = max(hm[c], hv);
            }
          }
        } else {
          size_t const pb = static_cast<size_t>(pr + lane) * d.rs + static_cast<size_t>(lane) * cw;
          i32 hprev = lane == 0 ? DP::c0(d.H0[pr])
                                : d.H[static_cast<size_t>(pr + lane - 1) * d.rs + static_cast<size_t>(lane) * cw - 1];
#pragma unroll
          for (int c = 0; c < CWM; ++c) {
            bool const in = static_cast<u32>(c) < cw;
            i32 const hc = in ? d.H[pb + c] : 0;
            i32 const fc = in ? d.F[pb + c] : 0;
            i32 const oc = in ? d.O[pb + c] : 0;
            i32 const mc = (nch == ((scp[c >> 2] >> (8 * (c & 3))) & 0xFFu)) ? M_ : N_;
            i32 const fv = max(hc + G_, fc + E_);
            i32 const ov = max(hc + Q_, oc + C_);
            i32 const hv = hprev + mc;
            hprev = hc;
            if (x == 0) {
              f[c] = fv;
              o[c] = ov;
              hm[c] = hv;
            } else {
              f[c] = max(f[c], fv);
              o[c] = max(o[c], ov);
              hm[c] = max(hm[c], hv);
            }
          }
        }
      }
      i32 hleft, eleft, qleft;
      if (lane == 0) {  // column 0 (kept in LDS): H = max(O, F), E = Q = -inf
        hleft = max(col0[2 * i], col0[2 * i + 1]);
        eleft = kNegInf;
        qleft = kNegInf;
      } else {
        hleft = hL;
        eleft = eL;
        qleft = qL;
      }
      pHl = hleft;
      i32 ev[CWM], qv[CWM];
#pragma unroll
      for (int c = 0; c < CWM; ++c) {
        i32 const e = max(hleft + G_, eleft + E_);
        i32 const q = max(hleft + Q_, qleft + C_);
        i32 const h = max(hm[c], max(max(f[c], e), max(o[c], q)));
        ev[c] = e;
        qv[c] = q;
        if (jb + c < je) {  // columns past the haplotype end are never read back
          hleft = h;
          eleft = e;
          qleft = q;
        }
        pH[c] = h;
        pF[c] = f[c];
        pO[c] = o[c];
      }
      size_t const ob = static_cast<size_t>(i + lane) * d.rs + static_cast<size_t>(lane) * cw;  // 8-element aligned
#pragma unroll
      for (int c8 = 0; c8 < CWM / 8; ++c8) {
        if (static_cast<u32>(c8 * 8) < cw) {
          int const c = c8 * 8;
          if (static_cast<u32>(c + 4) < cw) {
            *reinterpret_cast<uint4*>(d.H + ob + c) = make_uint4(sat_pack(pH[c], pH[c + 1]), sat_pack(pH[c + 2], pH[c + 3]), sat_pack(pH[c + 4], pH[c + 5]), sat_pack(pH[c + 6], pH[c + 7]));
            *reinterpret_cast<uint4*>(d.F + ob + c) = make_uint4(sat_pack(f[c], f[c + 1]), sat_pack(f[c + 2], f[c + 3]), sat_pack(f[c + 4], f[c + 5]), sat_pack(f[c + 6], f[c + 7]));
            *reinterpret_cast<uint4*>(d.O + ob + c) = make_uint4(sat_pack(o[c], o[c + 1]), sat_pack(o[c + 2], o[c + 3]), sat_pack(o[c + 4], o[c + 5]), sat_pack(o[c + 6], o[c + 7]));
            *reinterpret_cast<uint4*>(d.E + ob + c) = make_uint4(sat_pack(ev[c], ev[c + 1]), sat_pack(ev[c + 2], ev[c + 3]), sat_pack(ev[c + 4], ev[c + 5]), sat_pack(ev[c + 6], ev[c + 7]));
            *reinterpret_cast<uint4*>(d.Q + ob + c) = make_uint4(sat_pack(qv[c], qv[c + 1]), sat_pack(qv[c + 2], qv[c + 3]), sat_pack(qv[c + 4], qv[c + 5]), sat_pack(qv[c + 6], qv[c + 7]));
          } else {  // cw == c + 4: only four columns left in this lane's chunk
            *reinterpret_cast<uint2*>(d.H + ob + c) = make_uint2(sat_pack(pH[c], pH[c + 1]), sat_pack(pH[c + 2], pH[c + 3]));
            *reinterpret_cast<uint2*>(d.F + ob + c) = make_uint2(sat_pack(f[c], f[c + 1]), sat_pack(f[c + 2], f[c + 3]));
            *reinterpret_cast<uint2*>(d.O + ob + c) = make_uint2(sat_pack(o[c], o[c + 1]), sat_pack(o[c + 2], o[c + 3]));
            *reinterpret_cast<uint2*>(d.E + ob + c) = make_uint2(sat_pack(ev[c], ev[c + 1]), sat_pack(ev[c + 2], ev[c + 3]));
            *reinterpret_cast<uint2*>(d.Q + ob + c) = make_uint2(sat_pack(qv[c], qv[c + 1]), sat_pack(qv[c + 2], qv[c + 3]));
          }
        }
      }
      hl = hleft;
      el = eleft;
      ql = qleft;
    }
  }
  __threadfence_block();
}

// Generic path for haplotypes longer than 64 * CWM columns: same recurrences and layout, every
// predecessor row is read back from HBM (no register-resident row).
__device__ void poa_fill_long(const PG& g, const DP& d, u32 V, u32 L, int lane, const u8* seq, LdsArr<i32> col0) {
  u32 const cw = d.cw;
  u32 const jb = 1 + lane * cw;
  u32 const je = min(L + 1, jb + cw);
  u32 const nl = (L + cw - 1) / cw;
  i32 hl = 0, el = 0, ql = 0;
  for (u32 t = 0; t + 1 < V + nl; ++t) {
    i32 const hL = __shfl_up(hl, 1), eL = __shfl_up(el, 1), qL = __shfl_up(ql, 1);
    i32 const row = static_cast<i32>(t) - lane + 1;
    __threadfence_block();
    if (lane < static_cast<int>(nl) && row >= 1 && row <= static_cast<i32>(V)) {
      u32 const i = static_cast<u32>(row);
      u32 const node = g.rank2node[i - 1];
      u32 const np = g.nin[node];
      u8 const nch = g.nchar[node];
      i32 hleft, eleft, qleft;
      if (lane == 0) {
        hleft = max(col0[2 * i], col0[2 * i + 1]);
        eleft = kNegInf;
        qleft = kNegInf;
      } else {
        hleft = hL;
        eleft = eL;
        qleft = qL;
      }
      for (u32 j = jb; j < je; ++j) {
        i32 const mc = (nch == seq[j - 1]) ? M_ : N_;
        i32 f = kNegInf, o = kNegInf, hm = kNegInf;
        for (u32 x = 0; x < (np ? np : 1u); ++x) {
          u32 const pr = np ? row_pred(g, node, x) : 0u;
          i32 const hpj = d.h(pr, j);
          i32 const fv = max(hpj + G_, d.f(pr, j) + E_);
          i32 const ov = max(hpj + Q_, d.o(pr, j) + C_);
          i32 const hv = d.h(pr, j - 1) + mc;
          f = x == 0 ? fv : max(f, fv);
          o = x == 0 ? ov : max(o, ov);
          hm = x == 0 ? hv : max(hm, hv);
        }
        i32 const e = max(hleft + G_, eleft + E_);
        i32 const q = max(hleft + Q_, qleft + C_);
        i32 const h = max(hm, max(max(f, e), max(o, q)));
        size_t const ox = d.off(i, j);
        d.F[ox] = static_cast<i16>(DP::c0(f));
        d.O[ox] = static_cast<i16>(DP::c0(o));
        d.E[ox] = static_cast<i16>(DP::c0(e));
        d.Q[ox] = static_cast<i16>(DP::c0(q));
        d.H[ox] = static_cast<i16>(DP::c0(h));
        hleft = h;
        eleft = e;
        qleft = q;
      }
      hl = hleft;
      el = eleft;
      ql = qleft;
    }
  }
  __threadfence_block();
}

#ifdef MA_PROFILE
__device__ unsigned long long g_prof[16];
#define PROF_T0() unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define PROF_ACC(slot)                                                        \
  do {                                                                        \
    unsigned long long _t1 = __builtin_amdgcn_s_memtime();                    \
    if (lane == 0) atomicAdd(&g_prof[slot], _t1 - _t0);                       \
    _t0 = _t1;                                                                \
  } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF_ACC(slot) do {} while (0)
#endif

struct Shared {
  u32 V, L, W;
  u32 go;
  u32 abort_;
};

}  // namespace

struct MsaArgs {
  DBatch b;
  ma_asm_out_t a;
  ma_var_out_t o;
  PoaWs ws;
  ma_params_t prm;
  int win0;
};

__global__ __launch_bounds__(64) void k_msa(MsaArgs A) {
  __shared__ Shared sh;
  int const lw = blockIdx.x;
  int const w = A.win0 + lw;
  int const lane = threadIdx.x;
  ma_params_t const& P = A.prm;
  PoaWs const& ws = A.ws;
  int const MC = P.max_comps, MH = P.max_haps, ML = P.max_hap_len, MV = P.max_vars, MA = P.max_alts,
            MP = P.max_allele_bytes;

  u32 const ncomp = (A.a.win_status[w] & MA_W_NO_HAPLOTYPE) ? 0u : A.a.win_ncomp[w];
  if (ncomp == 0) {
    if (lane == 0) A.o.win_nvars[w] = 0;
    return;
  }
  // ---- carve the LDS graph ----
  u32 const PN = ws.pn;
  PG g;
  g.pn = PN;
  {
    u32 const b16 = 6 * PN;  // PN is a multiple of 8
    g.nchar.off = 0;
    g.nin.off = PN;
    g.nout.off = 2 * PN;
    g.nal.off = 3 * PN;
    g.marks.off = 4 * PN;
    g.ignored.off = 5 * PN;
    g.in_tail.off = b16;
    g.out_head.off = b16 + 2 * (kPE * PN);
    g.out_lab.off = b16 + 2 * (2 * kPE * PN);
    g.al.off = b16 + 2 * (3 * kPE * PN);
    g.rank2node.off = b16 + 2 * (4 * kPE * PN);
    g.node2rank.off = g.rank2node.off + 2 * PN;
    g.stack.off = g.node2rank.off + 2 * PN;  // 4 * PN + 8 u16 == 2 * (PN + 1) i32 (+ slack) for column 0
    g.stack_cap = 4 * PN;
  }
  LdsArr<i32> col0{g.stack.off};                       // [(PN + 1) * 2]: O, F of column 0
  LdsArr<u16> aln{g.stack.off + 2 * (4 * PN + 8)};    // [(PN + max_l + 2) * 2]
  u32 const aln_cap = PN + ws.max_l + 2;
  DP d;
  d.H = ws.H + static_cast<size_t>(lw) * ws.cells;
  d.F = ws.F + static_cast<size_t>(lw) * ws.cells;
  d.E = ws.E + static_cast<size_t>(lw) * ws.cells;
  d.O = ws.O + static_cast<size_t>(lw) * ws.cells;
  d.Q = ws.Q + static_cast<size_t>(lw) * ws.cells;
  {
    i32* c0 = ws.C0 + static_cast<size_t>(lw) * 5 * (PN + 2);
    d.H0 = c0;
    d.F0 = c0 + (PN + 2);
    d.E0 = c0 + 2 * (PN + 2);
    d.O0 = c0 + 3 * (PN + 2);
    d.Q0 = c0 + 4 * (PN + 2);
  }
  d.cw = 4;
  d.rs = 256;
  i16* H = d.H;  // also the raw-allele scratch of the bubble walk

  u32 nvars = 0, pool = 0;
  bool overflow = false;

  for (u32 c = 0; c < ncomp && !overflow; ++c) {
    size_t const ci = static_cast<size_t>(w) * MC + c;
    u32 const hap0 = A.a.comp_hap0[ci], nh = A.a.comp_nhaps[ci];
    if (lane == 0) {
      g.nn = g.nseq = g.nrank = 0;
      g.overflow = nh > 16;  // label masks are 16 bit
    }
    for (u32 h = 0; h < nh; ++h) {
      size_t const hi = static_cast<size_t>(w) * MH + hap0 + h;
      const u8* seq = A.a.hap_bases + hi * ML;
      u32 const L = A.a.hap_len[hi];
      PROF_T0();
      // ---- lane 0: column 0 of the DP (SisdAlignmentEngine::Initialize, kNW convex) ----
      if (lane == 0) {
        sh.go = 0;
        sh.abort_ = g.overflow ? 1u : 0u;
        if (!g.overflow && g.nn > 0 && L > 0) {
          u32 const V = g.nrank, W = L + 1;
          u32 const cwl = ((L + 63) / 64 + 7) & ~7u;
          if (static_cast<size_t>(V + 65) * 64 * cwl > ws.cells) {
            g.overflow = true;
            sh.abort_ = 1;
          } else {
            col0[0] = 0;
            col0[1] = 0;
            for (u32 i = 1; i <= V; ++i) {
              u32 const node = g.rank2node[i - 1];
              u32 const np = g.nin[node];
              i32 pen_o = np == 0 ? Q_ - C_ : kNegInf;
              i32 pen_f = np == 0 ? G_ - E_ : kNegInf;
              for (u32 x = 0; x < np; ++x) {
                u32 const pr = row_pred(g, node, x);
                pen_o = max(pen_o, col0[2 * pr]);
                pen_f = max(pen_f, col0[2 * pr + 1]);
              }
              col0[2 * i] = pen_o + C_;
              col0[2 * i + 1] = pen_f + E_;
            }
            sh.V = V;
            sh.L = L;
            sh.W = W;
            sh.go = 1;
          }
        }
      }
      __syncthreads();
      if (sh.go) {
        u32 const V = sh.V;
        d.cw = ((L + 63) / 64 + 7) & ~7u;
        d.rs = 64 * d.cw;
        // column 0 and row 0 to HBM (the traceback may touch them)
        for (u32 i = lane; i <= V; i += 64) {
          i32 const ov = col0[2 * i], fv = col0[2 * i + 1];
          d.O0[i] = ov;
          d.F0[i] = fv;
          d.Q0[i] = i == 0 ? 0 : kNegInf;
          d.E0[i] = i == 0 ? 0 : kNegInf;
          d.H0[i] = i == 0 ? 0 : max(ov, fv);
        }
        for (u32 j = 1 + lane; j <= L; j += 64) {
          size_t const ox = d.off(0, j);
          i32 const qv = Q_ + static_cast<i32>(j - 1) * C_, ev = G_ + static_cast<i32>(j - 1) * E_;
          d.O[ox] = -32768;
          d.Q[ox] = static_cast<i16>(DP::c0(qv));
          d.F[ox] = -32768;
          d.E[ox] = static_cast<i16>(DP::c0(ev));
          d.H[ox] = static_cast<i16>(DP::c0(max(qv, ev)));
        }
        __threadfence_block();
        __syncthreads();
        PROF_ACC(0);
        if (d.cw <= CWM) poa_fill(g, d, V, L, lane, seq, col0);
        else poa_fill_long(g, d, V, L, lane, seq, col0);
        __syncthreads();
        PROF_ACC(1);
      }
      // ---- lane 0: best end cell, traceback (SisdAlignmentEngine::Convex backtrack), graph update ----
      if (lane == 0 && !g.overflow) {
        u32 naln = 0;
        if (sh.go) {
          u32 const V = sh.V;
          i32 max_score = kNegInf;
          u32 max_i = 0, max_j = 0;
          for (u32 r = 0; r < V; ++r) {
            if (g.nout[g.rank2node[r]] != 0) continue;
            i32 const hv = d.h(r + 1, L);
            if (max_score < hv) {
              max_score = hv;
              max_i = r + 1;
              max_j = L;
            }
          }
          u32 i = max_i, j = max_j, prev_i = 0, prev_j = 0;
          auto push = [&](bool has_node, u32 node, bool has_pos, u32 pos) {
            if (naln + 1 >= aln_cap) {
              g.overflow = true;
              return;
            }
            aln[2 * naln] = has_node ? static_cast<u16>(node + 1) : 0;
            aln[2 * naln + 1] = has_pos ? static_cast<u16>(pos + 1) : 0;
            naln++;
          };
          while (!(i == 0 && j == 0) && !(max_i == 0 && max_j == 0) && !g.overflow) {
            i32 const Hij = d.h(i, j);
            bool found = false, ext_left = false, ext_up = false;
            u32 const node = i ? g.rank2node[i - 1] : 0u;
            u32 const np = i ? g.nin[node] : 0u;
            if (i != 0 && j != 0) {
              i32 const mc = (g.nchar[node] == seq[j - 1]) ? M_ : N_;
              for (u32 x = 0; x < (np ? np : 1u); ++x) {
                u32 const pi = np ? row_pred(g, node, x) : 0u;
                if (Hij == d.h(pi, j - 1) + mc) {
                  prev_i = pi;
                  prev_j = j - 1;
                  found = true;
                  break;
                }
              }
            }
            if (!found && i != 0) {
              for (u32 x = 0; x < (np ? np : 1u); ++x) {
                u32 const pi = np ? row_pred(g, node, x) : 0u;
                i32 const hpj = d.h(pi, j);
                bool ok = (ext_up |= (Hij == d.f(pi, j) + E_));
                if (!ok) ok = Hij == hpj + G_;
                if (!ok) ok = (ext_up |= (Hij == d.o(pi, j) + C_));
                if (!ok) ok = Hij == hpj + Q_;
                if (ok) {
                  prev_i = pi;
                  prev_j = j;
                  found = true;
                  break;
                }
              }
            }
            if (!found && j != 0) {
              i32 const hl1 = d.h(i, j - 1);
              bool ok = (ext_left |= (Hij == d.e(i, j - 1) + E_));
              if (!ok) ok = Hij == hl1 + G_;
              if (!ok) ok = (ext_left |= (Hij == d.q(i, j - 1) + C_));
              if (!ok) ok = Hij == hl1 + Q_;
              if (ok) {
                prev_i = i;
                prev_j = j - 1;
                found = true;
              }
            }
            push(i != prev_i, node, j != prev_j, j - 1);
            i = prev_i;
            j = prev_j;
            if (ext_left) {
              while (!g.overflow) {
                push(false, 0, true, j - 1);
                --j;
                bool const e_stop = d.e(i, j) + E_ != d.e(i, j + 1);
                bool const q_stop = d.q(i, j) + C_ != d.q(i, j + 1);
                if (e_stop && q_stop) break;
              }
            } else if (ext_up) {
              while (!g.overflow) {
                bool stop = true;
                prev_i = 0;
                u32 const nd2 = g.rank2node[i - 1];
                u32 const np2 = g.nin[nd2];
                i32 const fc = d.f(i, j), oc = d.o(i, j);
                for (u32 x = 0; x < np2; ++x) {
                  u32 const pr = row_pred(g, nd2, x);
                  if (fc == d.f(pr, j) + E_ || oc == d.o(pr, j) + C_) {
                    prev_i = pr;
                    stop = false;
                    break;
                  }
                }
                if (stop) {
                  for (u32 x = 0; x < np2; ++x) {
                    u32 const pr = row_pred(g, nd2, x);
                    i32 const hp2 = d.h(pr, j);
                    if (fc == hp2 + G_ || oc == hp2 + Q_) {
                      prev_i = pr;
                      break;
                    }
                  }
                }
                push(true, nd2, false, 0);
                i = prev_i;
                if (stop || i == 0) break;
              }
            }
          }
          // std::reverse(alignment)
          for (u32 x = 0; x < naln / 2; ++x) {
            u16 const a0 = aln[2 * x], a1 = aln[2 * x + 1];
            aln[2 * x] = aln[2 * (naln - 1 - x)];
            aln[2 * x + 1] = aln[2 * (naln - 1 - x) + 1];
            aln[2 * (naln - 1 - x)] = a0;
            aln[2 * (naln - 1 - x) + 1] = a1;
          }
        }
        PROF_ACC(2);
        if (!g.overflow) pg_add_alignment(g, aln, naln, seq, L);
        PROF_ACC(3);
      }
      __syncthreads();
    }

    // ---- lane 0: VariantExtractor over the component's POA graph ----
    if (lane == 0) {
      if (g.overflow) {
        overflow = true;
      } else if (g.nseq >= 2) {
        u32 const ns = g.nseq;
        i32 active[16];
        u32 hap_pos[16], starts[16];
        for (u32 s = 0; s < ns; ++s) {
          active[s] = g.seq_first[s];
          hap_pos[s] = 0;
        }
        u32 ref_pos = A.a.comp_anchor[ci];  // window-relative ref_anchor_pos (variant_builder.cpp:146)
        i32 prev_match = -1;
        u8* pl = A.o.allele_pool + static_cast<size_t>(w) * MP;
        // raw allele strings live in tmp memory: [ns][cap]
        u32 const acap = 2 * ws.max_l + 8;
        u8* raw = reinterpret_cast<u8*>(H);  // DP matrices are free now
        u32 rawlen[16];
        auto converged = [&]() {
          for (u32 s = 1; s < ns; ++s)
            if (active[s] != active[0]) return false;
          return true;
        };
        while (true) {
          if (converged()) {
            if (active[0] < 0) break;
            prev_match = active[0];
            for (u32 s = 0; s < ns; ++s)
              if (active[s] >= 0) {
                active[s] = pg_successor(g, static_cast<u32>(active[s]), s);
                hap_pos[s]++;
              }
            ref_pos++;
            continue;
          }
          bool const has_prev = prev_match >= 0;
          u32 const aoff = has_prev ? 1u : 0u;
          u32 start_pos = ref_pos - aoff;
          for (u32 s = 0; s < ns; ++s) {
            rawlen[s] = 0;
            if (has_prev) raw[s * acap + rawlen[s]++] = g.nchar[prev_match];
            starts[s] = hap_pos[s] - aoff;
          }
          while (!converged()) {
            u32 min_rank = 0xFFFFFFFFu;
            for (u32 s = 0; s < ns; ++s)
              if (active[s] >= 0) min_rank = min(min_rank, g.node2rank[active[s]]);
            if (min_rank == 0xFFFFFFFFu) break;
            for (u32 s = 0; s < ns; ++s)
              if (active[s] >= 0 && g.node2rank[active[s]] == min_rank) {
                if (rawlen[s] < acap) raw[s * acap + rawlen[s]++] = g.nchar[active[s]]; else overflow = true;
                active[s] = pg_successor(g, static_cast<u32>(active[s]), s);
                hap_pos[s]++;
                if (s == 0) ref_pos++;
              }
          }
          // group identical non-REF alleles (CreateNormalizedBubble); alt_of[s] = group id or -1
          i32 alt_of[16];
          u32 grp_rep[16];
          u32 ngrp = 0;
          for (u32 s = 1; s < ns; ++s) {
            alt_of[s] = -1;
            if (bytes_cmp(raw + s * acap, rawlen[s], raw, rawlen[0]) == 0) continue;
            for (u32 gi = 0; gi < ngrp; ++gi)
              if (bytes_cmp(raw + s * acap, rawlen[s], raw + grp_rep[gi] * acap, rawlen[grp_rep[gi]]) == 0) {
                alt_of[s] = static_cast<i32>(gi);
                break;
              }
            if (alt_of[s] < 0) {
              grp_rep[ngrp] = s;
              alt_of[s] = static_cast<i32>(ngrp++);
            }
          }
          if (ngrp == 0) continue;
          // NormalizeVcfParsimony (variant_bubble.cpp:89-116): trims act on views [lo, hi) of the raw strings
          u32 rlo = 0, rhi = rawlen[0];
          u32 glo[16], ghi[16];
          for (u32 gi = 0; gi < ngrp; ++gi) {
            glo[gi] = 0;
            ghi[gi] = rawlen[grp_rep[gi]];
          }
          if (rawlen[0] > 0) {
            while (rhi - rlo > 1) {  // right trim
              bool ok = true;
              for (u32 gi = 0; gi < ngrp && ok; ++gi)
                ok = (ghi[gi] - glo[gi] > 1) && raw[grp_rep[gi] * acap + ghi[gi] - 1] == raw[rhi - 1];
              if (!ok) break;
              rhi--;
              for (u32 gi = 0; gi < ngrp; ++gi) ghi[gi]--;
            }
            u32 const init_len = rhi - rlo;
            while (rhi - rlo > 1) {  // left trim
              bool ok = true;
              for (u32 gi = 0; gi < ngrp && ok; ++gi)
                ok = (ghi[gi] - glo[gi] > 1) && raw[grp_rep[gi] * acap + glo[gi]] == raw[rlo];
              if (!ok) break;
              rlo++;
              for (u32 gi = 0; gi < ngrp; ++gi) glo[gi]++;
            }
            start_pos += init_len - (rhi - rlo);
          }
          // AssembleMultiallelicVariant: ALTs sorted by sequence (variant_extractor.cpp:229)
          u32 ordg[16];
          for (u32 gi = 0; gi < ngrp; ++gi) {
            u32 jx = gi;
            while (jx > 0 && bytes_cmp(raw + grp_rep[ordg[jx - 1]] * acap + glo[ordg[jx - 1]], ghi[ordg[jx - 1]] - glo[ordg[jx - 1]],
                                       raw + grp_rep[gi] * acap + glo[gi], ghi[gi] - glo[gi]) > 0) {
              ordg[jx] = ordg[jx - 1];
              --jx;
            }
            ordg[jx] = gi;
          }
          u32 need = rhi - rlo;
          for (u32 gi = 0; gi < ngrp; ++gi) need += ghi[gi] - glo[gi];
          if (static_cast<int>(nvars) >= MV || static_cast<int>(ngrp) > MA || static_cast<int>(pool + need) > MP) {
            overflow = true;
            break;
          }
          size_t const vi = static_cast<size_t>(w) * MV + nvars;
          A.o.var_comp[vi] = c;
          A.o.var_pos[vi] = start_pos;
          A.o.var_ref_start[vi] = starts[0];
          A.o.var_ref_off[vi] = pool;
          A.o.var_ref_len[vi] = rhi - rlo;
          for (u32 x = rlo; x < rhi; ++x) pl[pool++] = raw[x];
          A.o.var_nalts[vi] = ngrp;
          for (int hx = 0; hx < MH; ++hx) {
            A.o.var_hap_allele[vi * MH + hx] = 0;
            A.o.var_hap_start[vi * MH + hx] = 0;
          }
          u32 rank_of_grp[16];
          for (u32 ai = 0; ai < ngrp; ++ai) {
            u32 const gi = ordg[ai];
            rank_of_grp[gi] = ai;
            const u8* as = raw + grp_rep[gi] * acap + glo[gi];
            u32 const alen = ghi[gi] - glo[gi];
            A.o.alt_off[vi * MA + ai] = pool;
            A.o.alt_len[vi * MA + ai] = alen;
            for (u32 x = 0; x < alen; ++x) pl[pool++] = as[x];
            const u8* rs = pl + A.o.var_ref_off[vi];
            i32 const ty = classify_variant(rs, rhi - rlo, as, alen);
            A.o.alt_type[vi * MA + ai] = ty;
            A.o.alt_length[vi * MA + ai] = variant_length(rs, rhi - rlo, as, alen, ty);
          }
          for (u32 s = 1; s < ns; ++s)
            if (alt_of[s] >= 0) {
              A.o.var_hap_allele[vi * MH + s] = static_cast<u8>(rank_of_grp[alt_of[s]] + 1);
              A.o.var_hap_start[vi * MH + s] = starts[s];
            }
          A.o.var_hap_start[vi * MH + 0] = starts[0];
          nvars++;
        }
      }
      sh.abort_ = overflow ? 1u : 0u;
    }
    __syncthreads();
    overflow = sh.abort_ != 0;
  }
  if (lane == 0) {
    A.o.win_nvars[w] = nvars;
    if (overflow) A.a.win_status[w] |= MA_W_VAR_OVERFLOW;
  }
}

#ifdef MA_PROFILE
extern "C" void ma_debug_prof(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
  }
}
#endif

int launch_msa(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o) {
  int const n = b.n_windows;
  if (n == 0) return MA_OK;
  ma_params_t const& P = ctx->prm;
  if (P.max_haps > 16) {
    ctx->err = "ma_msa_batch: max_haps > 16 not supported (16-bit haplotype label masks)";
    return MA_ERR_PARAM;
  }
  PoaWs ws{};
  // longest haplotype of the batch decides the DP width and the LDS graph capacity (one small D2H)
  u32 max_len = 0;
  {
    size_t const cnt = static_cast<size_t>(n) * P.max_haps;
    std::vector<u32> hl(cnt), st(n), nc(n), h0(static_cast<size_t>(n) * P.max_comps), nh(static_cast<size_t>(n) * P.max_comps);
    MA_HIP(ctx, hipMemcpyAsync(hl.data(), a.hap_len, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, hipMemcpyAsync(st.data(), a.win_status, 4ull * n, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, hipMemcpyAsync(nc.data(), a.win_ncomp, 4ull * n, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, hipMemcpyAsync(h0.data(), a.comp_hap0, 4ull * n * P.max_comps, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, hipMemcpyAsync(nh.data(), a.comp_nhaps, 4ull * n * P.max_comps, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int w = 0; w < n; ++w) {
      if (st[w] & MA_W_NO_HAPLOTYPE) continue;
      for (u32 c = 0; c < nc[w]; ++c) {
        size_t const ci = static_cast<size_t>(w) * P.max_comps + c;
        for (u32 h = 0; h < nh[ci]; ++h) max_len = std::max(max_len, hl[static_cast<size_t>(w) * P.max_haps + h0[ci] + h]);
      }
    }
  }
  max_len = std::max<u32>(max_len, 16);
  ws.max_l = max_len;
  u32 pn = max_len + std::max<u32>(256, max_len / 4);
  if (const char* e = getenv("MA_POA_NODE_CAP")) pn = static_cast<u32>(atoi(e));
  pn = std::min<u32>((pn + 7) & ~7u, 65000);
  // LDS: 6 B + 16 * 2 B + 2 * 2 B + 4 * 2 B per node, + alignment path
  auto lds_bytes = [&](u32 p) { return size_t(6) * p + size_t(2) * (4 * kPE * p + 2 * p + 4 * p + 8) + size_t(4) * (p + max_len + 2) + 64; };
  while (lds_bytes(pn) > 156 * 1024 && pn > max_len + 32) pn -= 8;
  ws.pn = pn;
  size_t const lds = lds_bytes(pn);
  ws.cw_max = ((max_len + 63) / 64 + 7) & ~7u;
  ws.cells = static_cast<size_t>(pn + 66) * 64 * ws.cw_max;
  if (ws.cells < static_cast<size_t>(P.max_haps) * (2 * max_len + 8) / 2 + 64)  // raw-allele scratch lives in H
    ws.cells = static_cast<size_t>(P.max_haps) * (2 * max_len + 8) / 2 + 64;

  size_t const per_window = 5 * ws.cells * 2 + 5 * (static_cast<size_t>(pn) + 2) * 4 + 8192;
  size_t budget = size_t(24) << 30;
  {
    size_t free_b = 0, total_b = 0;  // size the in-flight window count for the GPU's HBM (288 GB on MI355X)
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      budget = static_cast<size_t>(static_cast<double>(free_b + ctx->ws_poa.cap) * 0.55);
  }
  if (const char* e = getenv("MA_WS_GB")) budget = static_cast<size_t>(atoi(e)) << 30;
  int const chunk = static_cast<int>(std::max<size_t>(1, std::min<size_t>(n, budget / per_window)));
  MA_HIP(ctx, ctx->ws_poa.reserve(per_window * static_cast<size_t>(chunk)));
  if (lds > 65536)
    MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_msa), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
  for (int win0 = 0; win0 < n; win0 += chunk) {
    int const nwin = std::min(chunk, n - win0);
    char* base = static_cast<char*>(ctx->ws_poa.p);
    size_t const msz = (static_cast<size_t>(nwin) * ws.cells * 2 + 255) & ~size_t(255);
    ws.H = reinterpret_cast<i16*>(base);
    ws.F = reinterpret_cast<i16*>(base + msz);
    ws.E = reinterpret_cast<i16*>(base + 2 * msz);
    ws.O = reinterpret_cast<i16*>(base + 3 * msz);
    ws.Q = reinterpret_cast<i16*>(base + 4 * msz);
    ws.C0 = reinterpret_cast<i32*>(base + 5 * msz);
    MsaArgs args{b, a, o, ws, P, win0};
    ctx->tic("k_msa");
    hipLaunchKernelGGL(k_msa, dim3(nwin), dim3(64), lds, ctx->stream, args);
    ctx->toc();
    MA_HIP(ctx, hipGetLastError());
  }
  return MA_OK;
}

}  // namespace ma
