// Haplotype <-> reference partial-order alignment + variant extraction on gfx950.
//
// Replaces caller::MsaBuilder::UpdateSpoaState (caller/msa_builder.cpp:29-42; SPOA 4.1.5 engine
// kNW with convex gaps 0/-6/-6,-2/-26,-1: caller/msa_builder.h:72-89) and the caller::VariantSet
// constructor (caller/variant_extractor.cpp:24-233, variant_bubble.cpp:16-116, raw_variant.cpp:44-77).
//
// One wavefront per window; the window's components are processed one after another.  For every
// haplotype the sequence-to-DAG DP (five i32 matrices H,F,E,O,Q kept in HBM because SPOA's traceback
// compares VALUES) is filled by all 64 lanes as a skewed pipeline: lane l owns a contiguous chunk of
// columns and handles graph row (t - l) at step t, receiving the (H,E,Q) of the column to its left
// from lane l-1 through a wave shuffle; predecessor rows are read back from HBM.  Traceback, graph
// update (Graph::AddAlignment), the DFS topological sort and the bubble walk are short serial loops
// driven by lane 0.  Edge weights are not tracked: the reference only reads topology, labels and
// ranks from the POA graph (variant_extractor.cpp:47-58, :84-94, :159-181).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ma_internal.h"

namespace ma {

namespace {

constexpr int kPIn = 8;    // in/out edges per POA node
constexpr int kPAl = 4;    // aligned nodes per POA node
constexpr i32 kNegInf = static_cast<i32>(0x80000000u) + 1024;
constexpr i32 M_ = 0, N_ = -6, G_ = -6, E_ = -2, Q_ = -26, C_ = -1;  // msa_builder.h:72-77

struct PoaWs {
  u32 pnc;        // node capacity
  u32 pec;        // edge capacity
  size_t cells;   // DP cells per matrix per window
  u32 max_l;
  u8* nchar;      // [w][pnc]
  u8* nin;
  u8* nout;
  u8* nal;
  u32* in_e;      // [w][pnc][kPIn] edge ids
  u32* out_e;
  u32* al;        // [w][pnc][kPAl] node ids
  u32* e_tail;    // [w][pec]
  u32* e_head;
  u32* e_lab;     // label bitmask
  u32* rank2node; // [w][pnc]
  u32* node2rank; // [w][pnc]
  u32* row_pred0; // [w][pnc+2] offset into preds
  u32* preds;     // [w][pec]  predecessor ROW indices (rank + 1) in in-edge order
  u32* tmp;       // [w][5*pnc] marks / ignored / dfs stack
  i32* aln;       // [w][2*(pnc + max_l + 2)] alignment pairs (node id | -1, seq pos | -1)
  i32* H;         // [w][cells]
  i32* F;
  i32* E;
  i32* O;
  i32* Q;
};

struct PG {  // per-window graph view (lane 0 mutates it)
  u32 nn, ne, nseq, nrank;
  u32 pnc, pec;
  u8 *nchar, *nin, *nout, *nal;
  u32 *in_e, *out_e, *al, *e_tail, *e_head, *e_lab, *rank2node, *node2rank;
  i32 seq_first[32];
  bool overflow;
};

__device__ i32 pg_add_node(PG& g, u8 ch) {
  if (g.nn >= g.pnc) {
    g.overflow = true;
    return 0;
  }
  u32 const id = g.nn++;
  g.nchar[id] = ch;
  g.nin[id] = g.nout[id] = g.nal[id] = 0;
  return static_cast<i32>(id);
}
__device__ void pg_add_edge(PG& g, u32 tail, u32 head) {  // spoa::Graph::AddEdge (weights dropped)
  u32 const label = g.nseq;
  for (int x = 0; x < g.nout[tail]; ++x) {
    u32 const ei = g.out_e[tail * kPIn + x];
    if (g.e_head[ei] == head) {
      g.e_lab[ei] |= (1u << label);
      return;
    }
  }
  if (g.ne >= g.pec || g.nout[tail] >= kPIn || g.nin[head] >= kPIn) {
    g.overflow = true;
    return;
  }
  u32 const ei = g.ne++;
  g.e_tail[ei] = tail;
  g.e_head[ei] = head;
  g.e_lab[ei] = 1u << label;
  g.out_e[tail * kPIn + g.nout[tail]++] = ei;
  g.in_e[head * kPIn + g.nin[head]++] = ei;
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
  for (int x = 0; x < g.nout[node]; ++x) {
    u32 const ei = g.out_e[node * kPIn + x];
    if (g.e_lab[ei] & (1u << label)) return static_cast<i32>(g.e_head[ei]);
  }
  return -1;
}
// spoa::Graph::TopologicalSort
__device__ void pg_toposort(PG& g, u32* tmp) {
  u8* marks = reinterpret_cast<u8*>(tmp);
  u8* ignored = reinterpret_cast<u8*>(tmp + g.pnc / 4 + 1);
  u32* stack = tmp + 2 * (g.pnc / 4 + 1);
  u32 const stack_cap = 4 * g.pnc;
  for (u32 i = 0; i < g.nn; ++i) marks[i] = ignored[i] = 0;
  g.nrank = 0;
  for (u32 s = 0; s < g.nn; ++s) {
    if (marks[s] != 0) continue;
    u32 sp = 0;
    stack[sp++] = s;
    while (sp > 0) {
      u32 const cur = stack[sp - 1];
      bool valid = true;
      if (marks[cur] != 2) {
        for (int x = 0; x < g.nin[cur]; ++x) {
          u32 const t = g.e_tail[g.in_e[cur * kPIn + x]];
          if (marks[t] != 2) {
            if (sp < stack_cap) stack[sp++] = t; else g.overflow = true;
            valid = false;
          }
        }
        if (!ignored[cur]) {
          for (int x = 0; x < g.nal[cur]; ++x) {
            u32 const an = g.al[cur * kPAl + x];
            if (marks[an] != 2) {
              if (sp < stack_cap) stack[sp++] = an; else g.overflow = true;
              ignored[an] = 1;
              valid = false;
            }
          }
        }
        if (valid) {
          marks[cur] = 2;
          if (!ignored[cur]) {
            g.rank2node[g.nrank++] = cur;
            for (int x = 0; x < g.nal[cur]; ++x) g.rank2node[g.nrank++] = g.al[cur * kPAl + x];
          }
        } else {
          marks[cur] = 1;
        }
      }
      if (valid) sp--;
      if (g.overflow) return;
    }
  }
  for (u32 r = 0; r < g.nrank; ++r) g.node2rank[g.rank2node[r]] = r;
}

// spoa::Graph::AddAlignment
__device__ void pg_add_alignment(PG& g, const i32* aln, u32 naln, const u8* seq, u32 len, u32* tmp) {
  if (len == 0) return;
  if (naln == 0) {
    i32 const first = pg_add_sequence(g, seq, 0, len);
    g.seq_first[g.nseq++] = first;
    if (!g.overflow) pg_toposort(g, tmp);
    return;
  }
  i32 vfront = -1, vback = -1;
  for (u32 x = 0; x < naln; ++x)
    if (aln[2 * x + 1] != -1) {
      if (vfront < 0) vfront = aln[2 * x + 1];
      vback = aln[2 * x + 1];
    }
  i32 begin = pg_add_sequence(g, seq, 0, static_cast<u32>(vfront));
  i32 prev = begin >= 0 ? static_cast<i32>(g.nn - 1) : -1;
  i32 const last = pg_add_sequence(g, seq, static_cast<u32>(vback) + 1, len);
  for (u32 x = 0; x < naln && !g.overflow; ++x) {
    i32 const sp = aln[2 * x + 1];
    if (sp == -1) continue;
    u8 const ch = seq[sp];
    i32 curr = -1;
    i32 const nd = aln[2 * x];
    if (nd == -1) {
      curr = pg_add_node(g, ch);
    } else {
      u32 const jt = static_cast<u32>(nd);
      if (g.nchar[jt] == ch) {
        curr = nd;
      } else {
        for (int y = 0; y < g.nal[jt]; ++y)
          if (g.nchar[g.al[jt * kPAl + y]] == ch) {
            curr = static_cast<i32>(g.al[jt * kPAl + y]);
            break;
          }
        if (curr < 0) {
          curr = pg_add_node(g, ch);
          if (g.overflow) break;
          int const na = g.nal[jt];
          if (na + 1 > kPAl) {
            g.overflow = true;
            break;
          }
          for (int y = 0; y < na; ++y) {
            u32 const kt = g.al[jt * kPAl + y];
            g.al[kt * kPAl + g.nal[kt]++] = static_cast<u32>(curr);
            g.al[static_cast<u32>(curr) * kPAl + g.nal[curr]++] = kt;
          }
          g.al[jt * kPAl + g.nal[jt]++] = static_cast<u32>(curr);
          g.al[static_cast<u32>(curr) * kPAl + g.nal[curr]++] = jt;
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
  if (!g.overflow) pg_toposort(g, tmp);
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

struct Shared {
  u32 V, L, W;
  u32 go;        // 1: run the pipelined fill for the current haplotype
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
  size_t const nb = static_cast<size_t>(lw) * ws.pnc, eb = static_cast<size_t>(lw) * ws.pec;
  PG g;
  g.pnc = ws.pnc;
  g.pec = ws.pec;
  g.nchar = ws.nchar + nb;
  g.nin = ws.nin + nb;
  g.nout = ws.nout + nb;
  g.nal = ws.nal + nb;
  g.in_e = ws.in_e + nb * kPIn;
  g.out_e = ws.out_e + nb * kPIn;
  g.al = ws.al + nb * kPAl;
  g.e_tail = ws.e_tail + eb;
  g.e_head = ws.e_head + eb;
  g.e_lab = ws.e_lab + eb;
  g.rank2node = ws.rank2node + nb;
  g.node2rank = ws.node2rank + nb;
  u32* row_pred0 = ws.row_pred0 + static_cast<size_t>(lw) * (ws.pnc + 2);
  u32* preds = ws.preds + eb;
  u32* tmp = ws.tmp + nb * 5;
  i32* aln = ws.aln + static_cast<size_t>(lw) * 2 * (ws.pnc + ws.max_l + 2);
  i32* H = ws.H + static_cast<size_t>(lw) * ws.cells;
  i32* F = ws.F + static_cast<size_t>(lw) * ws.cells;
  i32* E = ws.E + static_cast<size_t>(lw) * ws.cells;
  i32* O = ws.O + static_cast<size_t>(lw) * ws.cells;
  i32* Q = ws.Q + static_cast<size_t>(lw) * ws.cells;

  u32 nvars = 0, pool = 0;
  bool overflow = false;

  for (u32 c = 0; c < ncomp && !overflow; ++c) {
    size_t const ci = static_cast<size_t>(w) * MC + c;
    u32 const hap0 = A.a.comp_hap0[ci], nh = A.a.comp_nhaps[ci];
    if (lane == 0) {
      g.nn = g.ne = g.nseq = g.nrank = 0;
      g.overflow = false;
    }
    for (u32 h = 0; h < nh; ++h) {
      size_t const hi = static_cast<size_t>(w) * MH + hap0 + h;
      const u8* seq = A.a.hap_bases + hi * ML;
      u32 const L = A.a.hap_len[hi];
      // ---- lane 0: row metadata + column 0 (SisdAlignmentEngine::Initialize, kNW convex) ----
      if (lane == 0) {
        sh.go = 0;
        sh.abort_ = g.overflow ? 1u : 0u;
        if (!g.overflow && g.nn > 0 && L > 0) {
          u32 const V = g.nrank, W = L + 1;
          if (static_cast<size_t>(V + 1) * W > ws.cells) {
            g.overflow = true;
            sh.abort_ = 1;
          } else {
            u32 po = 0;
            row_pred0[0] = 0;
            row_pred0[1] = 0;
            for (u32 r = 0; r < V; ++r) {
              u32 const nd = g.rank2node[r];
              for (int x = 0; x < g.nin[nd]; ++x) preds[po++] = g.node2rank[g.e_tail[g.in_e[nd * kPIn + x]]] + 1;
              row_pred0[r + 2] = po;
            }
            O[0] = 0; Q[0] = 0; F[0] = 0; E[0] = 0; H[0] = 0;
            for (u32 i = 1; i <= V; ++i) {
              u32 const p0 = row_pred0[i], p1 = row_pred0[i + 1];
              i32 pen_o = p0 == p1 ? Q_ - C_ : kNegInf;
              i32 pen_f = p0 == p1 ? G_ - E_ : kNegInf;
              for (u32 x = p0; x < p1; ++x) {
                pen_o = max(pen_o, O[static_cast<size_t>(preds[x]) * W]);
                pen_f = max(pen_f, F[static_cast<size_t>(preds[x]) * W]);
              }
              size_t const ix = static_cast<size_t>(i) * W;
              O[ix] = pen_o + C_;
              Q[ix] = kNegInf;
              F[ix] = pen_f + E_;
              E[ix] = kNegInf;
              H[ix] = max(O[ix], F[ix]);
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
        u32 const V = sh.V, W = sh.W;
        // row 0
        for (u32 j = 1 + lane; j <= L; j += 64) {
          O[j] = kNegInf;
          Q[j] = Q_ + static_cast<i32>(j - 1) * C_;
          F[j] = kNegInf;
          E[j] = G_ + static_cast<i32>(j - 1) * E_;
          H[j] = max(Q[j], E[j]);
        }
        __threadfence_block();
        __syncthreads();
        // ---- skewed pipeline fill (SisdAlignmentEngine::Convex, alignment phase) ----
        u32 const cw = (L + 63) / 64;                  // columns per lane
        u32 const jb = 1 + lane * cw;                  // first column of this lane
        u32 const je = min(L + 1, jb + cw);            // one past the last column
        u32 const nl = (L + cw - 1) / cw;              // lanes with a non-empty chunk
        i32 hl = 0, el = 0, ql = 0;                    // (H,E,Q) at my last column of the row just done
        for (u32 t = 0; t < V + nl - 1 + 1; ++t) {
          i32 const hL = __shfl_up(hl, 1), eL = __shfl_up(el, 1), qL = __shfl_up(ql, 1);
          i32 const row = static_cast<i32>(t) - lane + 1;
          if (lane < static_cast<int>(nl) && row >= 1 && row <= static_cast<i32>(V)) {
            u32 const i = static_cast<u32>(row);
            size_t const ix = static_cast<size_t>(i) * W;
            u8 const nch = g.nchar[g.rank2node[i - 1]];
            u32 const p0 = row_pred0[i], p1 = row_pred0[i + 1];
            i32 hleft, eleft, qleft;
            if (lane == 0) {
              hleft = H[ix];
              eleft = E[ix];
              qleft = Q[ix];
            } else {
              hleft = hL;
              eleft = eL;
              qleft = qL;
            }
            for (u32 j = jb; j < je; ++j) {
              i32 const mc = (nch == seq[j - 1]) ? M_ : N_;
              i32 f, o, hm;
              {
                size_t const px = static_cast<size_t>(p0 == p1 ? 0u : preds[p0]) * W;
                f = max(H[px + j] + G_, F[px + j] + E_);
                o = max(H[px + j] + Q_, O[px + j] + C_);
                hm = H[px + j - 1] + mc;
              }
              for (u32 x = p0 + 1; x < p1; ++x) {
                size_t const px = static_cast<size_t>(preds[x]) * W;
                f = max(f, max(H[px + j] + G_, F[px + j] + E_));
                o = max(o, max(H[px + j] + Q_, O[px + j] + C_));
                hm = max(hm, H[px + j - 1] + mc);
              }
              i32 const e = max(hleft + G_, eleft + E_);
              i32 const q = max(hleft + Q_, qleft + C_);
              i32 const h = max(hm, max(max(f, e), max(o, q)));
              F[ix + j] = f;
              O[ix + j] = o;
              E[ix + j] = e;
              Q[ix + j] = q;
              H[ix + j] = h;
              hleft = h;
              eleft = e;
              qleft = q;
            }
            hl = hleft;
            el = eleft;
            ql = qleft;
          }
          __threadfence_block();
        }
        __syncthreads();
      }
      // ---- lane 0: best end cell, traceback (SisdAlignmentEngine::Convex backtrack), graph update ----
      if (lane == 0 && !g.overflow) {
        u32 naln = 0;
        if (sh.go) {
          u32 const V = sh.V, W = sh.W;
          i32 max_score = kNegInf;
          u32 max_i = 0, max_j = 0;
          for (u32 r = 0; r < V; ++r) {
            if (g.nout[g.rank2node[r]] != 0) continue;
            i32 const hv = H[static_cast<size_t>(r + 1) * W + L];
            if (max_score < hv) {
              max_score = hv;
              max_i = r + 1;
              max_j = L;
            }
          }
          u32 i = max_i, j = max_j, prev_i = 0, prev_j = 0;
          u32 const aln_cap = ws.pnc + ws.max_l + 2;
          while (!(i == 0 && j == 0) && !(max_i == 0 && max_j == 0)) {
            size_t const ix = static_cast<size_t>(i) * W;
            i32 const Hij = H[ix + j];
            bool found = false, ext_left = false, ext_up = false;
            if (i != 0 && j != 0) {
              u8 const nch = g.nchar[g.rank2node[i - 1]];
              i32 const mc = (nch == seq[j - 1]) ? M_ : N_;
              u32 const p0 = row_pred0[i], p1 = row_pred0[i + 1];
              u32 const np = p1 - p0;
              for (u32 x = 0; x < (np ? np : 1u); ++x) {
                u32 const pi = np ? preds[p0 + x] : 0u;
                if (Hij == H[static_cast<size_t>(pi) * W + (j - 1)] + mc) {
                  prev_i = pi;
                  prev_j = j - 1;
                  found = true;
                  break;
                }
              }
            }
            if (!found && i != 0) {
              u32 const p0 = row_pred0[i], p1 = row_pred0[i + 1];
              u32 const np = p1 - p0;
              for (u32 x = 0; x < (np ? np : 1u); ++x) {
                u32 const pi = np ? preds[p0 + x] : 0u;
                size_t const px = static_cast<size_t>(pi) * W + j;
                bool ok = (ext_up |= (Hij == F[px] + E_));
                if (!ok) ok = Hij == H[px] + G_;
                if (!ok) ok = (ext_up |= (Hij == O[px] + C_));
                if (!ok) ok = Hij == H[px] + Q_;
                if (ok) {
                  prev_i = pi;
                  prev_j = j;
                  found = true;
                  break;
                }
              }
            }
            if (!found && j != 0) {
              bool ok = (ext_left |= (Hij == E[ix + j - 1] + E_));
              if (!ok) ok = Hij == H[ix + j - 1] + G_;
              if (!ok) ok = (ext_left |= (Hij == Q[ix + j - 1] + C_));
              if (!ok) ok = Hij == H[ix + j - 1] + Q_;
              if (ok) {
                prev_i = i;
                prev_j = j - 1;
                found = true;
              }
            }
            if (naln + 2 >= aln_cap) {
              g.overflow = true;
              break;
            }
            aln[2 * naln] = (i == prev_i) ? -1 : static_cast<i32>(g.rank2node[i - 1]);
            aln[2 * naln + 1] = (j == prev_j) ? -1 : static_cast<i32>(j - 1);
            naln++;
            i = prev_i;
            j = prev_j;
            if (ext_left) {
              while (true) {
                if (naln + 2 >= aln_cap) {
                  g.overflow = true;
                  break;
                }
                aln[2 * naln] = -1;
                aln[2 * naln + 1] = static_cast<i32>(j - 1);
                naln++;
                --j;
                size_t const rx = static_cast<size_t>(i) * W;
                bool const e_stop = E[rx + j] + E_ != E[rx + j + 1];
                bool const q_stop = Q[rx + j] + C_ != Q[rx + j + 1];
                if (e_stop && q_stop) break;
              }
            } else if (ext_up) {
              while (true) {
                bool stop = true;
                prev_i = 0;
                u32 const p0 = row_pred0[i], p1 = row_pred0[i + 1];
                size_t const cx = static_cast<size_t>(i) * W + j;
                for (u32 x = p0; x < p1; ++x) {
                  size_t const px = static_cast<size_t>(preds[x]) * W + j;
                  if (F[cx] == F[px] + E_ || O[cx] == O[px] + C_) {
                    prev_i = preds[x];
                    stop = false;
                    break;
                  }
                }
                if (stop) {
                  for (u32 x = p0; x < p1; ++x) {
                    size_t const px = static_cast<size_t>(preds[x]) * W + j;
                    if (F[cx] == H[px] + G_ || O[cx] == H[px] + Q_) {
                      prev_i = preds[x];
                      break;
                    }
                  }
                }
                if (naln + 2 >= aln_cap) {
                  g.overflow = true;
                  break;
                }
                aln[2 * naln] = static_cast<i32>(g.rank2node[i - 1]);
                aln[2 * naln + 1] = -1;
                naln++;
                i = prev_i;
                if (stop || i == 0) break;
              }
            }
            if (g.overflow) break;
          }
          // std::reverse(alignment)
          for (u32 x = 0; x < naln / 2; ++x) {
            i32 const a0 = aln[2 * x], a1 = aln[2 * x + 1];
            aln[2 * x] = aln[2 * (naln - 1 - x)];
            aln[2 * x + 1] = aln[2 * (naln - 1 - x) + 1];
            aln[2 * (naln - 1 - x)] = a0;
            aln[2 * (naln - 1 - x) + 1] = a1;
          }
        }
        if (!g.overflow) pg_add_alignment(g, aln, naln, seq, L, tmp);
      }
      __syncthreads();
    }

    // ---- lane 0: VariantExtractor over the component's POA graph ----
    if (lane == 0) {
      if (g.overflow) {
        overflow = true;
      } else if (g.nseq >= 2) {
        u32 const ns = g.nseq;
        i32 active[32];
        u32 hap_pos[32], starts[32];
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
        u32 rawlen[32];
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
          i32 alt_of[32];
          u32 grp_rep[32];
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
          u32 glo[32], ghi[32];
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
          u32 ordg[32];
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
          u32 rank_of_grp[32];
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

int launch_msa(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o) {
  int const n = b.n_windows;
  if (n == 0) return MA_OK;
  ma_params_t const& P = ctx->prm;
  PoaWs ws{};
  ws.max_l = static_cast<u32>(P.max_hap_len);
  ws.pnc = static_cast<u32>(2 * P.max_hap_len + 512);
  if (const char* e = getenv("MA_POA_NODE_CAP")) ws.pnc = static_cast<u32>(atoi(e));
  ws.pec = 2 * ws.pnc;
  // DP rows are bounded by the node capacity; columns by the longest haplotype.  To keep the footprint
  // proportional to the data, size the matrices from the batch's longest haplotype (one tiny D2H).
  u32 max_len = 0;
  {
    // hap_len is [n * max_haps]; a host-side max over a device array needs a copy
    size_t const cnt = static_cast<size_t>(n) * P.max_haps;
    std::vector<u32> hl(cnt);
    MA_HIP(ctx, hipMemcpyAsync(hl.data(), a.hap_len, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<u32> st(n), nc(n), h0(static_cast<size_t>(n) * P.max_comps), nh(static_cast<size_t>(n) * P.max_comps);
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
  u32 const rows_cap = std::min<u32>(ws.pnc, 2 * max_len + 256) + 1;
  ws.cells = static_cast<size_t>(rows_cap) * (max_len + 1);
  if (ws.cells < static_cast<size_t>(P.max_haps) * (2 * max_len + 8) / 4 + 64)  // raw-allele scratch lives in H
    ws.cells = static_cast<size_t>(P.max_haps) * (2 * max_len + 8) / 4 + 64;

  auto carve = [&](char* base, size_t A, PoaWs& g) -> size_t {
    size_t off = 0;
    auto take = [&](size_t bytes) {
      off = (off + 255) & ~size_t(255);
      char* p = base ? base + off : nullptr;
      off += bytes;
      return p;
    };
    size_t const NC = g.pnc, EC = g.pec;
    g.nchar = reinterpret_cast<u8*>(take(A * NC));
    g.nin = reinterpret_cast<u8*>(take(A * NC));
    g.nout = reinterpret_cast<u8*>(take(A * NC));
    g.nal = reinterpret_cast<u8*>(take(A * NC));
    g.in_e = reinterpret_cast<u32*>(take(A * NC * kPIn * 4));
    g.out_e = reinterpret_cast<u32*>(take(A * NC * kPIn * 4));
    g.al = reinterpret_cast<u32*>(take(A * NC * kPAl * 4));
    g.e_tail = reinterpret_cast<u32*>(take(A * EC * 4));
    g.e_head = reinterpret_cast<u32*>(take(A * EC * 4));
    g.e_lab = reinterpret_cast<u32*>(take(A * EC * 4));
    g.rank2node = reinterpret_cast<u32*>(take(A * NC * 4));
    g.node2rank = reinterpret_cast<u32*>(take(A * NC * 4));
    g.row_pred0 = reinterpret_cast<u32*>(take(A * (NC + 2) * 4));
    g.preds = reinterpret_cast<u32*>(take(A * EC * 4));
    g.tmp = reinterpret_cast<u32*>(take(A * NC * 5 * 4));
    g.aln = reinterpret_cast<i32*>(take(A * 2 * (NC + g.max_l + 2) * 4));
    g.H = reinterpret_cast<i32*>(take(A * g.cells * 4));
    g.F = reinterpret_cast<i32*>(take(A * g.cells * 4));
    g.E = reinterpret_cast<i32*>(take(A * g.cells * 4));
    g.O = reinterpret_cast<i32*>(take(A * g.cells * 4));
    g.Q = reinterpret_cast<i32*>(take(A * g.cells * 4));
    return off;
  };
  PoaWs probe = ws;
  size_t const per_window = carve(nullptr, 1, probe) + 4096;
  size_t budget = size_t(24) << 30;
  if (const char* e = getenv("MA_WS_GB")) budget = static_cast<size_t>(atoi(e)) << 30;
  int const chunk = static_cast<int>(std::max<size_t>(1, std::min<size_t>(n, budget / per_window)));
  MA_HIP(ctx, ctx->ws_poa.reserve(per_window * static_cast<size_t>(chunk)));
  for (int win0 = 0; win0 < n; win0 += chunk) {
    int const nwin = std::min(chunk, n - win0);
    carve(static_cast<char*>(ctx->ws_poa.p), static_cast<size_t>(nwin), ws);
    MsaArgs args{b, a, o, ws, P, win0};
    ctx->tic("k_msa");
    hipLaunchKernelGGL(k_msa, dim3(nwin), dim3(64), 0, ctx->stream, args);
    ctx->toc();
    MA_HIP(ctx, hipGetLastError());
  }
  return MA_OK;
}

}  // namespace ma
