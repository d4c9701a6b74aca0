// Haplotype <-> reference partial-order alignment + variant extraction on gfx950.
//
// Replaces caller::MsaBuilder::UpdateSpoaState (caller/msa_builder.cpp:29-42; SPOA 4.1.5 engine
// kNW with convex gaps 0/-6/-6,-2/-26,-1: caller/msa_builder.h:72-89) and the caller::VariantSet
// constructor (caller/variant_extractor.cpp:24-233, variant_bubble.cpp:16-116, raw_variant.cpp:44-77).
//
// One persistent kernel per chunk of windows (launch_msa -> k_poa, round 6), two kinds of jobs scheduled on the device:
//  * G jobs (msa_window) -- one workgroup of 256 threads (4 wavefronts) works on a window; its components and haplotypes are
//    processed one after another.  The POA graph (node chars, in/out adjacency with 16-bit ids, per-edge haplotype label
//    masks, aligned-node rings, rank <-> node maps) lives in LDS (~42 B per node).  msa_window is a coroutine: when the next
//    haplotype <-> graph alignment is ready for the DP it writes the row descriptors, saves its LDS block to HBM and
//    returns; the workgroup that takes the window after its fill restores the block and continues (best end cell,
//    certificate, traceback, graph update, toposort, variants).
//  * F jobs (band_job) -- ONE wavefront fills a band of 64 / 128 / 256 columns around every row's backbone coordinate
//    (poa_fill_lean / poa_fill_band), reading the descriptors straight from the saved block.  The band is exact by
//    certificate (see poa_fill_band) or the alignment is redone with the next tier, at last with the full fill.
//  (k_msa / k_msa_band run the same two bodies as rounds the host counts: MA_POA_SCHED=0, the scheduling of rounds 3-5.)
//  * Full fill (poa_fill<CW>, fallback and MA_POA_BAND=0): row-synchronous, all 256 lanes work on ONE DP row, lane l
//    owns CW consecutive columns; the horizontal gap chains (E, Q) are closed with two prefix maxima over the row
//    (DPP scans + one LDS exchange across the wave boundaries); the previous two rows stay in registers.  A row is
//    written to HBM only if a later row needs it and cannot take it from registers (a predecessor that is neither
//    rank-1 nor rank-2: the ends of indel arcs).
//  * SPOA's traceback compares VALUES of five i32 matrices.  Every comparison it can make at a cell
//    only involves values the fill has in registers when it computes that cell, so the fill evaluates
//    them on the spot and stores a 10-bit decision code per cell (2 B instead of 20 B of matrices):
//      [1:0] move: 0 diagonal, 1 up, 2 left          [2] the move opens an extension run (ext_up/ext_left)
//      [3]   left-extension run continues INTO this cell from its left neighbour
//      [4]   up-extension run continues through an F/O predecessor  [5] ... stops at an H predecessor
//      [7:6] predecessor index of the move             [9:8] predecessor index of the up-extension
//    The traceback reads one code per step; runs of diagonal moves over rank-consecutive rows, left runs
//    and up runs are detected 64 cells at a time with a wave ballot.
//  * Graph::AddAlignment is lane-parallel (every path node is touched by exactly one alignment entry);
//    SPOA's DFS topological sort ranks the trivial roots 64 at a time and runs the DFS only for the rest
//    (its order defines the ranks).
//  * The bubble walk of VariantExtractor skips converged stretches 256 nodes at a time.
// Edge weights are not tracked: the reference only reads topology, labels and ranks from the POA
// graph (variant_extractor.cpp:47-58, :84-94, :159-181).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ma_internal.h"

namespace ma {

namespace {

constexpr int kPE = 4;        // in / out edges and aligned nodes kept per POA node
constexpr int kT = 256;       // threads per window
constexpr u32 kSlowCap = 256; // rows with a cached predecessor-row list
constexpr u32 kMaxSeq = 32;   // haplotypes per component: the per-edge label masks are 16 bit (common kernel) or 32 bit (LAB32)
constexpr i32 kNegInf = static_cast<i32>(0x80000000u) + 1024;
constexpr i32 M_ = 0, N_ = -6, G_ = -6, E_ = -2, Q_ = -26, C_ = -1;  // msa_builder.h:72-77

// The band tier k_msa can fill on its own wave 0 (tail of a batch, MA_POA_BAND=1); every tier has its k_msa_band.
#ifndef MA_POA_TIER0
#define MA_POA_TIER0 2
#endif
constexpr u32 kTierIn = MA_POA_TIER0;

constexpr u32 RI_FAST = 1u << 11;     // single predecessor == previous rank
constexpr u32 RI_SLOWTAB = 1u << 12;  // predecessor rows cached in slowpred[info >> 16]
constexpr u32 RI_STORE = 1u << 13;    // a later row reads this row back from HBM
constexpr u32 RI_LEAN = 1u << 14;     // band tiers: the row takes the straight-line path of poa_fill_lean (see band_flags)
constexpr u32 RI_SLIDE = 1u << 15;    // ... and its window sits one lane to the right of the previous row's

struct PoaWs {
  u32 pn;            // node capacity (LDS)
  u32 max_l;         // longest haplotype of the batch
  u32 w_stride;      // i32 per stored row and matrix
  u32 row_slots;     // stored rows per window
  u32 use_band;      // try the banded fills first (see launch_msa: MA_POA_BAND)
  u32 tier0;         // columns per lane of the first band tier: 1 / 2 / 4 = 64 / 128 / 256 columns (MA_POA_TIER0)
  u32 no_wide_start; // (A/B) every alignment starts at tier0, whatever its length
  u32* tier_stats;   // [8] fills per tier 64/128/256 + (at 4) failed certificates per tier; null unless MA_VERBOSE
  unsigned long long* dstats;  // ma_timing_control mode 3 (else null): [7] band cells, [8] band fills, [9] full-fill cells,
                               // [10] alignments, [11] closed-form alignments (ma_internal.h: dev_stats)
  u32* pending_ctr;  // split mode: windows that yielded in the current k_msa launch
  u32 no_direct;     // MA_POA_NO_DIRECT: every alignment goes through a fill (tests: the shortcut changes nothing)
  u32 raw_cap;       // MA_POA_RAW_CAP: bytes per haplotype of the bubble walk's LDS scratch (tests: forces the HBM route)
  u32 lean;          // band tiers run poa_fill_lean (default) / poa_fill_band (MA_POA_LEAN=0; same codes, tested)
  u32 lab32;         // this pass runs the kernels with 32-bit label masks (components of 17 .. 32 haplotypes)
  u32 pass;          // 0: every window; 1: only windows whose components all hold <= 16 haplotypes; 2: only the others
  size_t code_cells; // bytes per decision-code plane of a window's OWN area: band fills ((pn + 2) rows x 256 columns at most)
  size_t row_cells;  // i32 per window: row_slots stored rows x 3 matrices x band_stride
  // The full row-synchronous fill (the last tier: an alignment whose 256-column band failed its certificate) needs whole
  // rows -- (pn + 2) x (max_len + 16) code bytes per plane, row_slots x 3 x w_stride stored cells: 10 MB per window of
  // 1 kb haplotypes where the band tiers need 2.7 -- and begins and ends inside one G job: its areas belong to the
  // WORKGROUP that runs it (k_poa: one per resident workgroup; the host-counted rounds: one per window, as before).
  u16* fcodes;
  i32* frows;
  size_t fcode_cells;   // bytes per plane
  size_t frow_cells;    // i32
  u32 full_by_worker;   // 1: fcodes / frows are indexed by blockIdx.x, 0: by the window
  u32 band_stride;      // i32 per stored row and matrix in the band fills (256: the widest tier)
  u16* codes;
  i32* rows;
  i32* hlast;        // [w][pn + 8] H(i, L)
  // split mode (banded fill in its own kernel, k_msa_band): image of a window's LDS between kernel rounds
  u32 split;
  u32 img_words;     // u32 per image (the whole dynamic LDS block of k_msa)
  u8* img;           // [w][img_words * 4]
};

// All per-window state lives in the kernel's dynamic LDS and is addressed as offsets into the
// __shared__ array (NOT through generic pointers kept in a struct: those compile to flat_load).
extern __shared__ __attribute__((aligned(16))) unsigned char ma_lds[];
template <class T>
struct LdsArr {
  u32 off;  // byte offset into ma_lds
  __device__ __forceinline__ T& operator[](u32 i) const { return *reinterpret_cast<T*>(&ma_lds[off + i * sizeof(T)]); }
};

struct WgState {
  u32 nn, nseq, nrank, overflow;
  i32 seq_first[kMaxSeq];
  u32 mode, V, L, cw, naln, nslow;
  u32 band, band_fail;  // band tier of this alignment's current attempt (columns per lane, 0 = full fill) / its traceback ran off the band
  u32 filled;           // split mode: k_msa_band has filled the pending alignment at tier `band`
  i32 edge_max;         // maximum H over the band's exit cells
  i32 edge0;            // ... the part known before the fill: column 0 of the rows whose window does not start at column 1
  i32 best;
  u32 best_row;
  i32 ftot[2][4][2];  // fill: per-wave totals of the two prefix maxima, double buffered by row parity
  i32 fbnd[2][4][4];  // fill: what the next wave's first lane needs to rebuild this wave's last column
  u32 wsum[4];
  i32 red_v[4];
  u32 red_r[4];
  i32 begin, prev0, lastn;
  u32 nv;
  u32 x_done, x_a, x_ns;
  u32 win_overflow;
  // split mode: where k_msa resumes after k_msa_band has filled the pending alignment, and thread 0's output cursor
  u32 c_cur, h_cur, pending, done;
  u32 nvars, pool, var_overflow;
  // thread 0's per-haplotype state of the bubble walk: dynamically indexed arrays, which as locals live in scratch memory
  // (an HBM round trip per access of a serial walk)
  i32 xa_active[kMaxSeq], xa_alt_of[kMaxSeq];
  u32 xa_hap_pos[kMaxSeq], xa_starts[kMaxSeq], xa_rawlen[kMaxSeq], xa_grp_rep[kMaxSeq], xa_glo[kMaxSeq], xa_ghi[kMaxSeq], xa_ordg[kMaxSeq],
      xa_rank_of_grp[kMaxSeq];
};
#define ST (*reinterpret_cast<WgState*>(ma_lds))
constexpr u32 kStBytes = (sizeof(WgState) + 15u) & ~15u;

struct GL {  // LDS layout of one window's POA graph + scratch
  u32 pn;
  LdsArr<u8> nchar, nin, nout, nal;
  LdsArr<u16> in_tail, out_head, al, rank2node, node2rank;
  // per-edge haplotype label masks: u16 (components of <= 16 haplotypes: the common kernel) or u32 (LAB32 kernels: <= 32).
  // lab32 is a template constant of the kernel that carves the view, so the accessors fold to one width.
  u32 lab32;
  u32 out_lab_off;
  __device__ __forceinline__ u32 lab(u32 i) const {
    return lab32 ? *reinterpret_cast<u32*>(&ma_lds[out_lab_off + 4u * i]) : static_cast<u32>(*reinterpret_cast<u16*>(&ma_lds[out_lab_off + 2u * i]));
  }
  __device__ __forceinline__ void lab_set(u32 i, u32 v) const {
    if (lab32) *reinterpret_cast<u32*>(&ma_lds[out_lab_off + 4u * i]) = v;
    else *reinterpret_cast<u16*>(&ma_lds[out_lab_off + 2u * i]) = static_cast<u16>(v);
  }
  __device__ __forceinline__ void lab_or(u32 i, u32 v) const { lab_set(i, lab(i) | v); }
  __device__ __forceinline__ u32 max_seq() const { return lab32 ? 32u : 16u; }
  LdsArr<u16> npos;  // backbone coordinate of a node (guides the DP band): reference index, or the neighbour's + 1
  // scratch, alignment phase
  LdsArr<u32> rowinfo;
  LdsArr<u16> rowslot, rowdepth, runhead, rowj0, slowpred, aln;
  u32 aln_cap;
  // scratch, graph update (aliases rowinfo .. rowdepth)
  LdsArr<u16> cnode, cpos, ccur;
  // scratch, topological sort (aliases everything above)
  LdsArr<u16> stack;
  LdsArr<u8> marks, ignored;
  u32 stack_cap;
  __device__ __forceinline__ u32 ubyte(const u8* p) const { return *p; }
  __device__ __forceinline__ u32 uword(const void* p) const { return *static_cast<const u32*>(p); }
};

__host__ __device__ inline size_t poa_lds_bytes(u32 pn, u32 ml, u32 lab32) {
  size_t const graph = size_t(lab32 ? 50 : 42) * pn;
  size_t const s_aln = size_t(12) * (pn + 2) + 8 * kSlowCap + size_t(4) * (pn + ml + 2);
  size_t const s_topo = size_t(10) * pn;
  return kStBytes + graph + (s_aln > s_topo ? s_aln : s_topo) + 16;
}

__device__ __forceinline__ GL poa_carve(u32 PN, u32 ML, u32 lab32) {
  GL g;
  g.pn = PN;
  g.lab32 = lab32;
  u32 o = kStBytes;
  g.nchar.off = o;
  g.nin.off = o + PN;
  g.nout.off = o + 2 * PN;
  g.nal.off = o + 3 * PN;
  o += 4 * PN;
  g.in_tail.off = o;
  g.out_head.off = o + 8 * PN;
  g.out_lab_off = o + 16 * PN;
  g.al.off = o + (lab32 ? 32 : 24) * PN;
  o += (lab32 ? 40 : 32) * PN;
  g.rank2node.off = o;
  g.node2rank.off = o + 2 * PN;
  g.npos.off = o + 4 * PN;
  o += 6 * PN;
  u32 const S = o;
  g.rowinfo.off = S;
  g.rowslot.off = S + 4 * (PN + 2);
  g.rowdepth.off = S + 6 * (PN + 2);
  g.runhead.off = S + 8 * (PN + 2);
  g.rowj0.off = S + 10 * (PN + 2);
  g.slowpred.off = S + 12 * (PN + 2);
  g.aln.off = S + 12 * (PN + 2) + 8 * kSlowCap;
  g.aln_cap = PN + ML + 2;
  g.cnode.off = S;
  g.cpos.off = S + 2 * (ML + 2);
  g.ccur.off = S + 4 * (ML + 2);
  g.stack.off = S;
  g.stack_cap = 4 * PN;
  g.marks.off = S + 8 * PN;
  g.ignored.off = S + 9 * PN;
  return g;
}

// Barrier that orders LDS traffic only: the pipeline's HBM stores stay in flight across steps.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- serial graph primitives (thread 0): spoa::Graph ----
__device__ __forceinline__ i32 pg_add_node(GL const& g, u8 ch, u32 pos) {
  if (ST.nn >= g.pn) {
    ST.overflow = 1;
    return 0;
  }
  u32 const id = ST.nn++;
  g.nchar[id] = ch;
  g.nin[id] = g.nout[id] = g.nal[id] = 0;
  g.npos[id] = static_cast<u16>(min(pos, 0xFFFFu));
  return static_cast<i32>(id);
}
// spoa::Graph::AddEdge (weights dropped).  Also used lane-parallel by the graph update: there every
// call touches the out-list of a distinct tail and the in-list of a distinct head.
__device__ __forceinline__ void pg_add_edge(GL const& g, u32 tail, u32 head, u32 lab) {
  u32 const no = g.nout[tail];
  for (u32 x = 0; x < no; ++x)
    if (g.out_head[tail * kPE + x] == head) {
      g.lab_or(tail * kPE + x, lab);
      return;
    }
  u32 const ni = g.nin[head];
  if (no >= kPE || ni >= kPE) {
    atomicOr(&ST.overflow, 1u);
    return;
  }
  g.out_head[tail * kPE + no] = static_cast<u16>(head);
  g.lab_set(tail * kPE + no, lab);
  g.nout[tail] = static_cast<u8>(no + 1);
  g.in_tail[head * kPE + ni] = static_cast<u16>(tail);
  g.nin[head] = static_cast<u8>(ni + 1);
}
__device__ __forceinline__ i32 pg_add_sequence(GL const& g, const u8* seq, u32 begin, u32 end) {  // spoa::Graph::AddSequence
  if (begin >= end) return -1;
  i32 prev = -1;
  u32 const first = ST.nn;
  u32 const lab = 1u << ST.nseq;
  for (u32 i = begin; i < end; ++i) {
    i32 const cur = pg_add_node(g, seq[i], i);  // unaligned stretch: its own index is as good a guess as any
    if (ST.overflow) return -1;
    if (prev >= 0) pg_add_edge(g, static_cast<u32>(prev), static_cast<u32>(cur), lab);
    prev = cur;
  }
  return static_cast<i32>(first);
}
__device__ __forceinline__ i32 pg_successor(GL const& g, u32 node, u32 label) {  // spoa::Graph::Node::Successor
  for (int x = 0; x < g.nout[node]; ++x)
    if (g.lab(node * kPE + x) & (1u << label)) return static_cast<i32>(g.out_head[node * kPE + x]);
  return -1;
}
// spoa::Graph::TopologicalSort, run by wave 0 (marks / ignored are zeroed by the caller).
// SPOA visits the roots s = 0, 1, 2, ... and runs a DFS over unranked ancestors from each unmarked one, so
// when root s is reached every node with a smaller id is already marked.  A root without aligned nodes
// whose in-neighbours all have smaller ids is therefore ranked on the spot; runs of such roots (and of
// already-marked ones) are handled 64 at a time with a ballot.  Every other root gets the serial DFS.
__device__ __forceinline__ void pg_toposort(GL const& g, int lane) {
  u32 const nn = ST.nn;
  u32 nrank = 0, s = 0;
  u32 overflow = 0;
  LdsArr<u16> const stack = g.stack;
  while (s < nn && !overflow) {
    u32 const v = s + static_cast<u32>(lane);
    bool trivial = false, marked = false;
    if (v < nn) {
      marked = g.marks[v] != 0;
      if (!marked && g.nal[v] == 0) {
        u32 const ni = g.nin[v];
        bool ok = true;
        for (u32 x = 0; x < ni; ++x) ok = ok && g.in_tail[v * kPE + x] < v;
        trivial = ok;
      }
    }
    unsigned long long const pm = __ballot(v < nn && (marked || trivial));
    u32 const run = pm == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~pm));
    if (run > 0) {
      unsigned long long const keep = run == 64 ? ~0ull : ((1ull << run) - 1ull);
      unsigned long long const tm = __ballot(trivial) & keep;
      if (static_cast<u32>(lane) < run && trivial) {
        u32 const r = nrank + static_cast<u32>(__popcll(tm & ((1ull << lane) - 1ull)));
        g.rank2node[r] = static_cast<u16>(v);
        g.marks[v] = 2;
      }
      nrank += static_cast<u32>(__popcll(tm));
      s += run;
      continue;
    }
    if (lane == 0) {  // root s is unmarked and not trivial
      u32 sp = 0;
      stack[sp++] = static_cast<u16>(s);
      while (sp > 0) {
        u32 const cur = stack[sp - 1];
        bool valid = true;
        if (g.marks[cur] != 2) {
          for (int x = 0; x < g.nin[cur]; ++x) {
            u32 const t = g.in_tail[cur * kPE + x];
            if (g.marks[t] != 2) {
              if (sp < g.stack_cap) stack[sp++] = static_cast<u16>(t); else overflow = 1;
              valid = false;
            }
          }
          if (!g.ignored[cur]) {
            for (int x = 0; x < g.nal[cur]; ++x) {
              u32 const an = g.al[cur * kPE + x];
              if (g.marks[an] != 2) {
                if (sp < g.stack_cap) stack[sp++] = static_cast<u16>(an); else overflow = 1;
                g.ignored[an] = 1;
                valid = false;
              }
            }
          }
          if (valid) {
            g.marks[cur] = 2;
            if (!g.ignored[cur]) {
              g.rank2node[nrank++] = static_cast<u16>(cur);
              for (int x = 0; x < g.nal[cur]; ++x) g.rank2node[nrank++] = g.al[cur * kPE + x];
            }
          } else {
            g.marks[cur] = 1;
          }
        }
        if (valid) sp--;
        if (overflow) break;
      }
    }
    nrank = __shfl(nrank, 0);
    overflow = __shfl(overflow, 0);
    s += 1;
  }
  if (lane == 0) {
    ST.nrank = nrank;
    if (overflow) ST.overflow = 1;
  }
}

__device__ __forceinline__ i32 classify_variant(const u8* r, u32 rl, const u8* a, u32 al) {  // raw_variant.cpp:44-77
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
__device__ __forceinline__ i32 variant_length(const u8* r, u32 rl, const u8* a, u32 al, i32 t) {  // variant_bubble.cpp:16-47
  if (t == 0) return 1;
  i32 const R = static_cast<i32>(rl), A = static_cast<i32>(al);
  if (t == 1 || t == 2 || t == 4) return A - R;
  i32 s = 0;
  while (s < R && s < A && r[s] == a[s]) s++;
  i32 e = 0;
  while (e < (R - s) && e < (A - s) && r[R - 1 - e] == a[A - 1 - e]) e++;
  return A - s - e;
}
__device__ __forceinline__ int bytes_cmp(const u8* a, u32 al, const u8* b, u32 bl) {  // std::string operator<=>
  u32 const m = al < bl ? al : bl;
  for (u32 i = 0; i < m; ++i)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return al < bl ? -1 : (al > bl ? 1 : 0);
}

// DP row of predecessor x of DP row i (row = rank + 1), in in-edge order
template <class G>
__device__ __forceinline__ u32 pred_row(G const& g, u32 i, u32 info, u32 x) {
  if (info & RI_FAST) return i - 1;
  if (info & RI_SLOWTAB) return g.slowpred[(info >> 16) * kPE + x];
  u32 const node = g.rank2node[i - 1];
  return static_cast<u32>(g.node2rank[g.in_tail[node * kPE + x]]) + 1u;
}

// closed forms of DP row 0 and DP column 0 (SisdAlignmentEngine::Initialize, kNW convex)
__device__ __forceinline__ i32 row0_h(u32 j) {
  return j == 0 ? 0 : max(Q_ + static_cast<i32>(j - 1) * C_, G_ + static_cast<i32>(j - 1) * E_);
}
__device__ __forceinline__ i32 col0_o(u32 d) { return Q_ + C_ * static_cast<i32>(d); }
__device__ __forceinline__ i32 col0_f(u32 d) { return G_ + E_ * static_cast<i32>(d); }
__device__ __forceinline__ i32 col0_h(u32 d) { return max(col0_o(d), col0_f(d)); }

#ifdef MA_PROFILE
__device__ unsigned long long g_prof2[16];  // k_poa: [0] ticks scheduling (thread 0), [1] G jobs, [2] F phases (workgroup), [3] idle, [4..7] F phases of
                                            // 1 / 2 / 3 / 4 fills, [8] G jobs, [9] idle polls, [10] whole kernel per workgroup
__device__ unsigned long long g_prof[16];
#define PROF_T0() unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define PROF_ACC(slot)                                                        \
  do {                                                                        \
    unsigned long long _t1 = __builtin_amdgcn_s_memtime();                    \
    if (tid == 0) atomicAdd(&g_prof[slot], _t1 - _t0);                        \
    _t0 = _t1;                                                                \
  } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF_ACC(slot) do {} while (0)
#endif

// block-wide exclusive scan of one value per thread (two barriers)
__device__ __forceinline__ u32 block_excl_scan(u32 v, int tid, u32& total) {
  int const lane = tid & 63, wave = tid >> 6;
  u32 inc = v;
  for (int d = 1; d < 64; d <<= 1) {
    u32 const y = __shfl_up(inc, d);
    if (lane >= d) inc += y;
  }
  if (lane == 63) ST.wsum[wave] = inc;
  __syncthreads();
  u32 base = 0, tot = 0;
  for (int k = 0; k < 4; ++k) {
    u32 const s = ST.wsum[k];
    if (k < wave) base += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

// block-wide exclusive prefix maximum of one value per thread (two barriers)
__device__ __forceinline__ u32 block_excl_scan_max(u32 v, int tid) {
  int const lane = tid & 63, wave = tid >> 6;
  u32 inc = v;
  for (int d = 1; d < 64; d <<= 1) {
    u32 const y = __shfl_up(inc, d);
    if (lane >= d) inc = max(inc, y);
  }
  if (lane == 63) ST.wsum[wave] = inc;
  u32 ex = __shfl_up(inc, 1);
  if (lane == 0) ex = 0;
  __syncthreads();
  for (int k = 0; k < 4; ++k)
    if (k < wave) ex = max(ex, ST.wsum[k]);
  __syncthreads();
  return ex;
}

// block-wide minimum of one value per thread (two barriers)
__device__ __forceinline__ u32 block_min(u32 v, int tid) {
  int const lane = tid & 63, wave = tid >> 6;
  for (int off = 32; off > 0; off >>= 1) v = min(v, static_cast<u32>(__shfl_xor(v, off)));
  if (lane == 0) ST.wsum[wave] = v;
  __syncthreads();
  u32 const r = min(min(ST.wsum[0], ST.wsum[1]), min(ST.wsum[2], ST.wsum[3]));
  __syncthreads();
  return r;
}

// ---- wave-wide prefix maximum with DPP (no LDS traffic) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ i32 dpp_mov(i32 x, i32 ident) {
  return __builtin_amdgcn_update_dpp(ident, x, CTRL, ROW_MASK, 0xF, false);
}
constexpr i32 kScanIdent = static_cast<i32>(0x80000000u);  // identity of max: lets the compiler fold mov_dpp + max
__device__ __forceinline__ i32 wave_incl_max(i32 x, i32 ident) {
  x = max(x, dpp_mov<0x111, 0xF>(x, ident));  // row_shr:1
  x = max(x, dpp_mov<0x112, 0xF>(x, ident));  // row_shr:2
  x = max(x, dpp_mov<0x114, 0xF>(x, ident));  // row_shr:4
  x = max(x, dpp_mov<0x118, 0xF>(x, ident));  // row_shr:8
  x = max(x, dpp_mov<0x142, 0xA>(x, ident));  // row_bcast:15 -> rows 1, 3
  x = max(x, dpp_mov<0x143, 0xC>(x, ident));  // row_bcast:31 -> rows 2, 3
  return x;
}
__device__ __forceinline__ i32 wave_shr1(i32 x, i32 ident) { return dpp_mov<0x138, 0xF>(x, ident); }  // wave_shr:1

// Decision codes live in two byte planes of one window's code area: plane A (bits 0-5 of the code: move, extension
// flags) is written for every row, plane B (bits 6-9: the predecessor indices) only for rows that are not FAST -- a FAST
// row has a single predecessor, index 0.  Half the bytes of one u16 per cell, and the traceback reads plane B only when
// it stands on such a row.
template <int CW>
__device__ __forceinline__ void store_code_bytes(u8* dst, const u32 (&cd)[CW], u32 shift) {
  if constexpr (CW % 4 == 0) {
#pragma unroll
    for (int c = 0; c < CW; c += 4)
      *reinterpret_cast<u32*>(dst + c) = ((cd[c] >> shift) & 0x3Fu) | (((cd[c + 1] >> shift) & 0x3Fu) << 8) |
                                         (((cd[c + 2] >> shift) & 0x3Fu) << 16) | (((cd[c + 3] >> shift) & 0x3Fu) << 24);
  } else if constexpr (CW % 2 == 0) {
#pragma unroll
    for (int c = 0; c < CW; c += 2)
      *reinterpret_cast<u16*>(dst + c) = static_cast<u16>(((cd[c] >> shift) & 0x3Fu) | (((cd[c + 1] >> shift) & 0x3Fu) << 8));
  } else {
#pragma unroll
    for (int c = 0; c < CW; ++c) dst[c] = static_cast<u8>((cd[c] >> shift) & 0x3Fu);
  }
}

// ---- the row-synchronous fill: decision codes + the rows later rows must read back ----
// All 256 lanes work on the same DP row; lane l owns the CW columns 1 + l CW .. .  The vertical and
// diagonal terms of a cell only need the predecessor rows (registers / row store).  The horizontal
// gap chains are closed with two prefix maxima over the row instead of a left-to-right sweep:
// with m_k = max(diagonal, F, O) of column k (m_0 = H of column 0) the recurrences
//     E_j = max(H_{j-1} + g, E_{j-1} + e)      Q_j = max(H_{j-1} + q, Q_{j-1} + c)      H_j = max(m_j, E_j, Q_j)
// unroll (all dropped terms are strictly dominated because g < e, q < c, q + 1 < 0) to
//     Q_j = q + (j-1) c + max_{k<j}(m_k - k c)
//     E_j = max( g + (j-1) e + max_{k<j}(m_k - k e),   Q_j-ish: q + g + (j-2) c + max_{k<j}(m_k - k c) )
// which are the SAME integers the sweep produces, so the backtrack tests see identical values.
// Rows have uniform shape across the workgroup, so rows with several / far predecessors cost one slower
// step instead of stalling a skewed pipeline.
// (always inlined, like everything that takes the graph view or the workspace by reference: an out-of-line callee needs
//  them in memory, and "memory" for a kernel argument is a private copy per thread -- half a KB of scratch stores by every
//  thread of every workgroup at kernel entry, 1 GB per launch, and a scratch load behind every LDS offset in the callee)
template <int CW>
__device__ __forceinline__ void poa_fill(GL const& g, PoaWs const& ws, u16* codes, i32* rows, i32* hlast, u32 V, u32 L, int tid,
                         const u8* seq) {
  static_assert(E_ == -2 && C_ == -1, "prefix-max keys are written for e = -2, c = -1");
  int const lane = tid & 63, wave = tid >> 6;
  u32 const gl = static_cast<u32>(tid);
  u32 const jb = 1 + gl * CW;
  u32 const je = min(L + 1, jb + CW);
  u32 const nl = (L + CW - 1) / CW;
  u32 const W = nl * CW;
  bool const lane_on = gl < nl;
  u32 const glL = (L - 1) / CW, cL = (L - 1) % CW;
  constexpr i32 NEG = kScanIdent;
  u32 sc[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) sc[c] = (lane_on && jb + c < je) ? seq[jb + c - 1] : 0u;
  constexpr bool kPrev2 = CW <= 8;  // wide lanes keep only one previous row in registers
  constexpr int CW2 = kPrev2 ? CW : 1;
  i32 H1[CW], F1[CW], O1[CW], H2[CW2], F2[CW2], O2[CW2];
#pragma unroll
  for (int c = 0; c < CW; ++c) H1[c] = F1[c] = O1[c] = 0;
#pragma unroll
  for (int c = 0; c < CW2; ++c) H2[c] = F2[c] = O2[c] = 0;
  i32 hl1 = 0, hl2 = 0;  // H(i-1, jb-1), H(i-2, jb-1)
  bool stored_prev = false;
#ifdef MA_PROFILE
  unsigned long long pa = 0, pb = 0, pc = 0, pn_gen = 0;
#endif
  u32 info = g.rowinfo[1];
  u32 depth = g.rowdepth[1];
  for (u32 i = 1; i <= V; ++i) {
#ifdef MA_PROFILE
    unsigned long long const q0 = __builtin_amdgcn_s_memtime();
#endif
    u32 const nch = info & 0xFFu, np = (info >> 8) & 7u;
    bool const fast = info & RI_FAST, store = info & RI_STORE;
    u32 const info_cur = info;
    i32 const h0 = col0_h(depth);
    if (i < V) {  // next row's descriptor: hide the LDS latency behind this row
      info = g.rowinfo[i + 1];
      depth = g.rowdepth[i + 1];
    }
    auto fetch = [&](u32 pr, i32(&th)[CW], i32(&tf)[CW], i32(&to)[CW], i32& thd) {
      if (pr == 0) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = row0_h(jb + c);
          tf[c] = kNegInf;
          to[c] = kNegInf;
        }
        thd = row0_h(jb - 1);
      } else if (pr + 1 == i) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = H1[c];
          tf[c] = F1[c];
          to[c] = O1[c];
        }
        thd = hl1;
      } else if (kPrev2 && pr + 2 == i) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = H2[kPrev2 ? c : 0];
          tf[c] = F2[kPrev2 ? c : 0];
          to[c] = O2[kPrev2 ? c : 0];
        }
        thd = hl2;
      } else {
        u32 const slot = g.rowslot[pr];
        const i32* rb = rows + static_cast<size_t>(slot) * 3 * ws.w_stride + static_cast<size_t>(gl) * CW;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          bool const in = lane_on && jb + c < je;
          th[c] = in ? rb[c] : 0;
          tf[c] = in ? rb[ws.w_stride + c] : 0;
          to[c] = in ? rb[2 * static_cast<size_t>(ws.w_stride) + c] : 0;
        }
        thd = gl == 0 ? col0_h(g.rowdepth[pr]) : (lane_on ? rb[-1] : 0);
      }
    };
    u32 const npe = np ? np : 1u;
    // ---- vertical + diagonal part: m = max(diag, F, O) ----
    i32 hh[CW], ff[CW], oo[CW], hmv[CW];
    if (fast) {
      i32 hd = hl1;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        i32 const ph = H1[c];
        ff[c] = max(F1[c] + E_, ph + G_);
        oo[c] = max(O1[c] + C_, ph + Q_);
        hmv[c] = hd + ((nch == sc[c]) ? M_ : N_);
        hd = ph;
      }
    } else {
      for (u32 x = 0; x < npe; ++x) {
        u32 const pr = np ? pred_row(g, i, info_cur, x) : 0u;
        i32 th[CW], tf[CW], to[CW], hd;
        fetch(pr, th, tf, to, hd);
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          i32 const fv = max(tf[c] + E_, th[c] + G_), ov = max(to[c] + C_, th[c] + Q_);
          i32 const hv = hd + ((nch == sc[c]) ? M_ : N_);
          hd = th[c];
          if (x == 0) {
            ff[c] = fv;
            oo[c] = ov;
            hmv[c] = hv;
          } else {
            ff[c] = max(ff[c], fv);
            oo[c] = max(oo[c], ov);
            hmv[c] = max(hmv[c], hv);
          }
        }
      }
    }
    // ---- prefix maxima of m_k + k and m_k + 2k over the row (column 0 enters through lane 0) ----
    i32 run1 = gl == 0 ? h0 : NEG, run2 = run1;
    i32 pl1 = run1, pl2 = run2;  // prefix BEFORE this lane's last column (for the next wave's first lane)
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      i32 const m = max(hmv[c], max(ff[c], oo[c]));
      hh[c] = m;
      i32 const j = static_cast<i32>(jb) + c;
      if (c == CW - 1) {
        pl1 = run1;
        pl2 = run2;
      }
      // no validity mask: columns past the haplotype end lie to the right of every valid column and
      // prefix maxima only flow rightwards
      run1 = max(run1, m + j);
      run2 = max(run2, m + 2 * j);
    }
    i32 const inc1 = wave_incl_max(run1, NEG), inc2 = wave_incl_max(run2, NEG);
    i32 const ex1 = wave_shr1(inc1, NEG), ex2 = wave_shr1(inc2, NEG);
    u32 const par = i & 1u;
    if (lane == 63) {
      ST.ftot[par][wave][0] = inc1;
      ST.ftot[par][wave][1] = inc2;
      ST.fbnd[par][wave][0] = max(ex1, pl1);
      ST.fbnd[par][wave][1] = max(ex2, pl2);
      ST.fbnd[par][wave][2] = hh[CW - 1];
    }
    if (stored_prev) __threadfence_block();  // last row's store is read back two or more rows from now
#ifdef MA_PROFILE
    unsigned long long const q1 = __builtin_amdgcn_s_memtime();
#endif
    lds_barrier();
#ifdef MA_PROFILE
    unsigned long long const q2 = __builtin_amdgcn_s_memtime();
#endif
    i32 cb1 = NEG, cb2 = NEG, cbp1 = NEG, cbp2 = NEG;  // all earlier waves / all waves before the previous one
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      i32 const t1 = ST.ftot[par][k][0], t2 = ST.ftot[par][k][1];
      cb1 = max(cb1, k < wave ? t1 : NEG);
      cb2 = max(cb2, k < wave ? t2 : NEG);
      cbp1 = max(cbp1, k + 1 < wave ? t1 : NEG);
      cbp2 = max(cbp2, k + 1 < wave ? t2 : NEG);
    }
    // ---- E, Q, H of this lane's columns ----
    i32 ee[CW], qq[CW];
    {
      i32 s1 = max(cb1, ex1), s2 = max(cb2, ex2);
      if (gl == 0) s1 = s2 = h0;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        i32 const j = static_cast<i32>(jb) + c;
        i32 const q = s1 + Q_ - (j - 1);
        i32 const e = max(s2 + G_ - 2 * (j - 1), s1 + Q_ + G_ - (j - 2));
        i32 const m = hh[c];
        s1 = max(s1, m + j);
        s2 = max(s2, m + 2 * j);
        ee[c] = e;
        qq[c] = q;
        hh[c] = max(m, max(e, q));
      }
    }
    // ---- (H, E, Q) of the column to the left of this lane ----
    i32 hN = wave_shr1(hh[CW - 1], 0), eN = wave_shr1(ee[CW - 1], 0), qN = wave_shr1(qq[CW - 1], 0);
    if (lane == 0) {
      if (wave == 0) {
        hN = h0;
        eN = kNegInf;
        qN = kNegInf;
      } else {  // rebuild the previous wave's last column from what its lane 63 published
        i32 const s1 = max(cbp1, ST.fbnd[par][wave - 1][0]), s2 = max(cbp2, ST.fbnd[par][wave - 1][1]);
        i32 const j = static_cast<i32>(jb) - 1;
        qN = s1 + Q_ - (j - 1);
        eN = max(s2 + G_ - 2 * (j - 1), s1 + Q_ + G_ - (j - 2));
        hN = max(ST.fbnd[par][wave - 1][2], max(eN, qN));
      }
    }
    // ---- decision codes ----
    u32 cd[CW];
    if (fast) {
      i32 hleft = hN, eleft = eN, qleft = qN;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        i32 const ph = H1[c];
        i32 const a1 = F1[c] + E_, a2 = ph + G_, a3 = O1[c] + C_, a4 = ph + Q_;
        i32 const fv = ff[c], ov = oo[c], hm = hmv[c], h = hh[c], e = ee[c];
        i32 const b1 = eleft + E_, b2 = hleft + G_, b3 = qleft + C_;
        // fv = max(a1, a2), ov = max(a3, a4), e = max(b1, b2), q = max(b3, b4) (the prefix-max values ARE these
        // maxima), so SPOA's ordered equality tests collapse to: which argument of the winning max is it?
        //   up:   h == a1 || (h != a2 && h == a3)   ==   (h == fv) ? a1 >= a2 : a3 >= a4      (given h == max(fv, ov))
        //   left: h == b1 || (h != b2 && h == b3)   ==   (h == e)  ? b1 >= b2 : b3 >= b4      (given h == max(e, q))
        i32 const b4 = hleft + Q_;
        bool const A1 = a1 >= a2, A3 = a3 >= a4, B1 = b1 >= b2, B3 = b3 >= b4;
        bool const D = h == hm, U = h == max(fv, ov);
        bool const eu = (h == fv) ? A1 : A3;
        bool const elx = (h == e) ? B1 : B3;
        bool const lc = B1 || B3;
        bool const us = A1 || A3;
        u32 code = D ? 0u : (U ? 1u : 2u);
        code |= (!D && (U ? eu : elx)) ? 4u : 0u;
        code |= lc ? 8u : 0u;
        code |= us ? 16u : 32u;
        cd[c] = code;
        hleft = h;
        eleft = e;
        qleft = qq[c];
      }
    } else {
      u32 elmask = 0, lcmask = 0;
      {
        i32 hleft = hN, eleft = eN, qleft = qN;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          i32 const h = hh[c];
          i32 const b1 = eleft + E_, b2 = hleft + G_, b3 = qleft + C_;
          if ((h == b1) || ((h != b2) && (h == b3))) elmask |= 1u << c;
          if ((b1 == ee[c]) || (b3 == qq[c])) lcmask |= 1u << c;
          hleft = h;
          eleft = ee[c];
          qleft = qq[c];
        }
      }
      // second pass over the predecessors: which one SPOA's backtrack would pick, in its test order
      u32 dmask = 0, umask = 0, eumask = 0, usmask = 0, uhmask = 0;
      u32 xs[CW];  // predecessor indices: [1:0] diagonal, [3:2] up, [5:4] F/O extension, [7:6] H extension
#pragma unroll
      for (int c = 0; c < CW; ++c) xs[c] = 0;
      for (u32 x = 0; x < npe; ++x) {
        u32 const pr = np ? pred_row(g, i, info_cur, x) : 0u;
        i32 th[CW], tf[CW], to[CW], hd;
        fetch(pr, th, tf, to, hd);
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          i32 const a1 = tf[c] + E_, a2 = th[c] + G_, a3 = to[c] + C_, a4 = th[c] + Q_;
          i32 const hv = hd + ((nch == sc[c]) ? M_ : N_);
          hd = th[c];
          i32 const h = hh[c];
          u32 const bit = 1u << c;
          if (!(dmask & bit) && h == hv) {
            dmask |= bit;
            xs[c] |= x;
          }
          bool const t1 = h == a1, t2 = h == a2, t3 = h == a3, t4 = h == a4;
          if (!(umask & bit) && (t1 || t2 || t3 || t4)) {
            umask |= bit;
            xs[c] |= x << 2;
            if (t1 || (!t2 && t3)) eumask |= bit;
          }
          if (np) {  // a row without in-edges has an empty predecessor loop in the up-extension walk
            if (!(usmask & bit) && ((ff[c] == a1) || (oo[c] == a3))) {
              usmask |= bit;
              xs[c] |= x << 4;
            }
            if (!(uhmask & bit) && ((ff[c] == a2) || (oo[c] == a4))) {
              uhmask |= bit;
              xs[c] |= x << 6;
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        u32 const bit = 1u << c;
        bool const D = dmask & bit, U = umask & bit;
        u32 code = D ? 0u : (U ? 1u : 2u);
        code |= (!D && (U ? ((eumask & bit) != 0) : ((elmask & bit) != 0))) ? 4u : 0u;
        code |= (lcmask & bit) ? 8u : 0u;
        bool const us = usmask & bit, uh = uhmask & bit;
        code |= us ? 16u : (uh ? 32u : 0u);
        code |= (D ? (xs[c] & 3u) : (U ? ((xs[c] >> 2) & 3u) : 0u)) << 6;
        code |= (us ? ((xs[c] >> 4) & 3u) : (uh ? ((xs[c] >> 6) & 3u) : 0u)) << 8;
        cd[c] = code;
      }
    }
    // the two previous rows stay in registers
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      if (kPrev2) {
        H2[kPrev2 ? c : 0] = H1[c];
        F2[kPrev2 ? c : 0] = F1[c];
        O2[kPrev2 ? c : 0] = O1[c];
      }
      H1[c] = hh[c];
      F1[c] = ff[c];
      O1[c] = oo[c];
    }
    hl2 = hl1;
    hl1 = hN;
    if (lane_on) {
      // decision codes, row-major byte planes: one store of the wave covers 64 * CW contiguous bytes
      u8* const cp = reinterpret_cast<u8*>(codes) + static_cast<size_t>(i) * W + static_cast<size_t>(gl) * CW;
      store_code_bytes<CW>(cp, cd, 0);
      if (!fast) store_code_bytes<CW>(cp + ws.fcode_cells, cd, 6);
      if (store) {
        u32 const slot = g.rowslot[i];
        i32* rb = rows + static_cast<size_t>(slot) * 3 * ws.w_stride + static_cast<size_t>(gl) * CW;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          if (jb + c < je) {
            rb[c] = hh[c];
            rb[ws.w_stride + c] = ff[c];
            rb[2 * static_cast<size_t>(ws.w_stride) + c] = oo[c];
          }
        }
      }
      if (gl == glL) {
        i32 v = hh[0];
#pragma unroll
        for (int c = 1; c < CW; ++c) v = (static_cast<u32>(c) == cL) ? hh[c] : v;
        hlast[i] = v;
      }
    }
    stored_prev = store;
#ifdef MA_PROFILE
    unsigned long long const q3 = __builtin_amdgcn_s_memtime();
    pa += q1 - q0;
    pb += q2 - q1;
    pc += q3 - q2;
    pn_gen += fast ? 0 : 1;
#endif
  }
#ifdef MA_PROFILE
  if (tid == 0) {
    atomicAdd(&g_prof[8], pa);
    atomicAdd(&g_prof[9], pb);
    atomicAdd(&g_prof[10], pc);
    atomicAdd(&g_prof[11], pn_gen);
    atomicAdd(&g_prof[12], static_cast<unsigned long long>(V));
  }
#endif
}

// ---- banded fill (wave 0 only): 256 columns around the expected diagonal, exact by certificate ----
// Haplotypes differ from the graph's backbone by a few variants, so the optimal path stays close to the
// diagonal "column = backbone coordinate of the row's node".  Row i gets the 256-column window
// [j0(i), j0(i) + 255] (j0 = 1 mod 4) centred on that coordinate and ONE wavefront fills it: no barriers, no
// LDS exchange, a quarter of the cells.
// Everything outside the window counts as minus infinity.  Exactness is certified afterwards:
//   every way a path can leave the band passes through an "exit" cell (a window cell whose up / diagonal /
//   right successor lies outside the successor row's window; row 0 / column 0 cells next to uncovered
//   ground); all transition scores are <= 0, so a path that ever left the band scores <= E = the maximum
//   H over exit cells.  If S (the best end score found inside the band) satisfies S - 32 > E then
//   (a) S is the true optimum, (b) every DP value the backtrack can test at a path cell is either one it
//   compares against a value >= S (then equality forces exactness on both sides) or a gap-state value that
//   is >= its path neighbour's H - 26 >= S - 26 > E, hence also attained inside the band: the banded
//   decision codes along the path equal the full DP's.  Otherwise the caller repeats the row-synchronous
//   full fill.
// Lane l owns the four window columns j0 + 4 l .. j0 + 4 l + 3.  When the window slides (always by a multiple
// of four columns, usually one lane every four rows) the register rows slide with it: a whole-wave DPP shift;
// the columns that fall off the left edge are exits, the ones that enter start at minus infinity.
__device__ __forceinline__ i32 wave_shl1(i32 x, i32 fill) { return dpp_mov<0x130, 0xF>(x, fill); }  // lane l <- lane l + 1
// the same, IN PLACE (the result is tied to x's own register, lane 63 is patched afterwards): written the other way the
// compiler gives the shifted rows new registers and pays 28 copies on every row that does NOT slide
__device__ __forceinline__ void wave_shl1_inplace(i32& x, i32 fill, bool last_lane) {
  x = __builtin_amdgcn_update_dpp(x, x, 0x130, 0xF, 0xF, false);
  x = last_lane ? fill : x;
}
__device__ __forceinline__ void wave_shr1_inplace(i32& x, i32 fill, bool first_lane) {  // lane l <- lane l - 1
  x = __builtin_amdgcn_update_dpp(x, x, 0x138, 0xF, 0xF, false);
  x = first_lane ? fill : x;
}

template <int CW, class G>
__device__ void poa_fill_band(G const& g, u32 const w_stride, size_t const plane, u16* codes, i32* rows, i32* hlast, u32 V,
                              u32 L, int lane, const u8* seq, i32* edge_out) {
  static_assert(CW == 1 || CW == 2 || CW == 4, "band tiers: 64, 128 or 256 columns");
  constexpr u32 BW = 64u * CW;  // columns of the band
  constexpr i32 NEG = kScanIdent;
  i32 jr[CW];  // this lane's columns relative to the window
#pragma unroll
  for (int c = 0; c < CW; ++c) jr[c] = CW * lane + c;
  i32 HA[CW], FA[CW], OA[CW], HB[CW], FB[CW], OB[CW];  // the two previous rows, roles alternating (see `row`)
#pragma unroll
  for (int c = 0; c < CW; ++c) HA[c] = FA[c] = OA[c] = HB[c] = FB[c] = OB[c] = kNegInf;
  u32 sc[CW];
  i32 edge = kNegInf;
  u32 info = g.rowinfo[1];
  u32 depth = g.rowdepth[1];
  u32 j0n = g.rowj0[1];
  u32 j0 = j0n;  // window the register rows are aligned to
  auto load_sc = [&]() {
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      u32 const j = j0 + static_cast<u32>(CW) * lane + c;
      sc[c] = (j >= 1 && j <= L) ? seq[j - 1] : 0u;
    }
  };
  load_sc();
  // the four characters that enter at lane 63 on the next one-lane slide (uniform; fetched one slide ahead)
  u32 sc_in[CW];
  auto load_sc_in = [&]() {
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      u32 const j = j0 + BW + c;
      sc_in[c] = j <= L ? g.ubyte(seq + (j - 1)) : 0u;
    }
  };
  load_sc_in();
  // one row: (H1, F1, O1) is row i - 1, (H2, F2, O2) row i - 2 -- which row i then OVERWRITES: the caller swaps the two
  // register sets from row to row instead of rotating them (24 copies a row, and as many again where the compiler
  // split the rotation across the loop edge)
  auto row = [&](u32 const i, i32(&H1)[CW], i32(&F1)[CW], i32(&O1)[CW], i32(&H2)[CW], i32(&F2)[CW], i32(&O2)[CW])
                 __attribute__((always_inline)) {
    u32 const nch = info & 0xFFu, np = (info >> 8) & 7u;
    bool const fast = info & RI_FAST, store = info & RI_STORE;
    u32 const info_cur = info;
    i32 const h0 = col0_h(depth);
    u32 const j0_new = j0n;
    if (i < V) {
      info = g.rowinfo[i + 1];
      depth = g.rowdepth[i + 1];
      j0n = g.rowj0[i + 1];
    }
    // ---- slide the register rows to this row's window ----
    if (__builtin_expect(j0_new != j0, 0)) {  // (one row in four: the register copies belong on this side)
      i32 const sh = (static_cast<i32>(j0_new) - static_cast<i32>(j0)) / CW;  // lanes; > 0: window moves right
      // columns that fall off: they had no successor inside the band
      {
        bool const dropped = sh > 0 ? lane < sh : lane >= 64 + sh;
        u32 const jbo = j0 + static_cast<u32>(CW) * lane;
        if (dropped) {
#pragma unroll
          for (int c = 0; c < CW; ++c)
            if (jbo + c >= 1 && jbo + c <= L) edge = max(edge, max(H1[c], H2[c]));
        }
      }
      // one lane at a time, every register IN PLACE (almost always a single step to the right); a second code path
      // that builds the shifted rows in new registers costs two dozen copies on every row that does not slide
      while (j0 < j0_new) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          wave_shl1_inplace(H1[c], kNegInf, lane == 63);
          wave_shl1_inplace(F1[c], kNegInf, lane == 63);
          wave_shl1_inplace(O1[c], kNegInf, lane == 63);
          wave_shl1_inplace(H2[c], kNegInf, lane == 63);
          wave_shl1_inplace(F2[c], kNegInf, lane == 63);
          wave_shl1_inplace(O2[c], kNegInf, lane == 63);
          i32 t = static_cast<i32>(sc[c]);
          wave_shl1_inplace(t, static_cast<i32>(sc_in[c]), lane == 63);
          sc[c] = static_cast<u32>(t);
        }
        j0 += CW;
        load_sc_in();
      }
      if (j0 > j0_new) {  // rare: the backbone coordinate steps back
        do {
          j0 -= CW;
#pragma unroll
          for (int c = 0; c < CW; ++c) {
            wave_shr1_inplace(H1[c], kNegInf, lane == 0);
            wave_shr1_inplace(F1[c], kNegInf, lane == 0);
            wave_shr1_inplace(O1[c], kNegInf, lane == 0);
            wave_shr1_inplace(H2[c], kNegInf, lane == 0);
            wave_shr1_inplace(F2[c], kNegInf, lane == 0);
            wave_shr1_inplace(O2[c], kNegInf, lane == 0);
            u32 const j = j0 + c;
            i32 t = static_cast<i32>(sc[c]);
            wave_shr1_inplace(t, (j >= 1 && j <= L) ? static_cast<i32>(seq[j - 1]) : 0, lane == 0);
            sc[c] = static_cast<u32>(t);
          }
        } while (j0 > j0_new);
        load_sc_in();
      }
    }
    u32 const jb = j0 + static_cast<u32>(CW) * lane;
    // ---- vertical + diagonal part ----
    i32 hh[CW], ff[CW], oo[CW], hmv[CW];
    // H(pr, jb - 1) from the left neighbour lane's last column, column 0 at the matrix edge, -inf at the band edge
    auto left_of = [&](i32 last_col_value, u32 pr) -> i32 {
      i32 const v = wave_shr1(last_col_value, kNegInf);
      if (lane != 0) return v;
      return jb == 1 ? (pr == 0 ? 0 : col0_h(g.rowdepth[pr])) : kNegInf;
    };
    // a stored row restricted to this window (its own window may sit elsewhere)
    auto fetch_store = [&](u32 pr, i32(&th)[CW], i32(&tf)[CW], i32(&to)[CW], i32& thd) {
      u32 const pj0 = g.rowj0[pr];
      int const d = (static_cast<i32>(j0) - static_cast<i32>(pj0)) / CW;  // this lane reads the stored lane l + d
      int const sl = lane + d;
      bool const in = sl >= 0 && sl < 64;
      u32 const slot = g.rowslot[pr];
      const i32* base = rows + static_cast<size_t>(slot) * 3 * w_stride;
#pragma unroll
      for (int c = 0; c < CW; ++c) th[c] = tf[c] = to[c] = kNegInf;
      if (in) {
        if constexpr (CW == 4) {
          int4 const vh = *reinterpret_cast<const int4*>(base + 4 * sl);
          int4 const vf = *reinterpret_cast<const int4*>(base + w_stride + 4 * sl);
          int4 const vo = *reinterpret_cast<const int4*>(base + 2 * static_cast<size_t>(w_stride) + 4 * sl);
          th[0] = vh.x; th[1] = vh.y; th[2] = vh.z; th[3] = vh.w;
          tf[0] = vf.x; tf[1] = vf.y; tf[2] = vf.z; tf[3] = vf.w;
          to[0] = vo.x; to[1] = vo.y; to[2] = vo.z; to[3] = vo.w;
        } else if constexpr (CW == 2) {
          int2 const vh = *reinterpret_cast<const int2*>(base + 2 * sl);
          int2 const vf = *reinterpret_cast<const int2*>(base + w_stride + 2 * sl);
          int2 const vo = *reinterpret_cast<const int2*>(base + 2 * static_cast<size_t>(w_stride) + 2 * sl);
          th[0] = vh.x; th[1] = vh.y;
          tf[0] = vf.x; tf[1] = vf.y;
          to[0] = vo.x; to[1] = vo.y;
        } else {
          th[0] = base[sl];
          tf[0] = base[w_stride + sl];
          to[0] = base[2 * static_cast<size_t>(w_stride) + sl];
        }
      }
      // H(pr, jb - 1): the stored column just left of this lane's first one
      i32 hd = kNegInf;
      if (jb == 1) {
        hd = col0_h(g.rowdepth[pr]);
      } else if (sl >= 1 && sl <= 64) {
        hd = base[CW * sl - 1];
      }
      thd = hd;
      // stored columns this window does not cover are exits of row pr towards this row
      if (d != 0) {
        int const ml = lane;  // stored lane inspected by this lane
        bool const dropped = d > 0 ? ml < d : ml >= 64 + d;
        if (dropped) {
          u32 const xj = pj0 + static_cast<u32>(CW) * ml;
#pragma unroll
          for (int c = 0; c < CW; ++c)
            if (xj + c <= L) edge = max(edge, base[CW * ml + c]);
        }
      }
    };
    auto fetch = [&](u32 pr, i32(&th)[CW], i32(&tf)[CW], i32(&to)[CW], i32& thd) {
      if (pr == 0) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = row0_h(jb + c);
          tf[c] = kNegInf;
          to[c] = kNegInf;
        }
        thd = row0_h(jb - 1);
      } else if (pr + 1 == i) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = H1[c];
          tf[c] = F1[c];
          to[c] = O1[c];
        }
        thd = left_of(H1[CW - 1], pr);
      } else if (pr + 2 == i) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = H2[c];
          tf[c] = F2[c];
          to[c] = O2[c];
        }
        thd = left_of(H2[CW - 1], pr);
      } else {
        fetch_store(pr, th, tf, to, thd);
      }
    };
    u32 const npe = np ? np : 1u;
    if (fast) {
      i32 hd = left_of(H1[CW - 1], i - 1);
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        i32 const ph = H1[c];
        ff[c] = max(F1[c] + E_, ph + G_);
        oo[c] = max(O1[c] + C_, ph + Q_);
        hmv[c] = hd + ((nch == sc[c]) ? M_ : N_);
        hd = ph;
      }
    } else {
      for (u32 x = 0; x < npe; ++x) {
        u32 const pr = np ? pred_row(g, i, info_cur, x) : 0u;
        i32 th[CW], tf[CW], to[CW], hd;
        fetch(pr, th, tf, to, hd);
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          i32 const fv = max(tf[c] + E_, th[c] + G_), ov = max(to[c] + C_, th[c] + Q_);
          i32 const hv = hd + ((nch == sc[c]) ? M_ : N_);
          hd = th[c];
          if (x == 0) {
            ff[c] = fv;
            oo[c] = ov;
            hmv[c] = hv;
          } else {
            ff[c] = max(ff[c], fv);
            oo[c] = max(oo[c], ov);
            hmv[c] = max(hmv[c], hv);
          }
        }
      }
      // row 0 ground that this window does not cover (a node without in-edges hangs off the virtual start row)
      if (np == 0 && j0 + BW <= L) edge = max(edge, row0_h(j0 + BW));
      if (np == 0 && j0 > 1) edge = max(edge, row0_h(1));
    }
    // the previous rows' cell whose diagonal successor would be the column right of this window
    if (lane == 63 && jb + CW - 1 < L) edge = max(edge, max(H1[CW - 1], H2[CW - 1]));
    // column 0 of this row is an exit when the window does not start at column 1
    if (j0 > 1) edge = max(edge, h0);
    // ---- prefix maxima over the window (column 0 enters through lane 0 when the window starts at column 1) ----
    // (columns beyond L are filled like any other: a cell only feeds cells to its right and below, so what they hold
    //  never reaches a column <= L; nothing reads them.  Column indices are taken RELATIVE to the window, jr = j - j0:
    //  the same maxima up to a per-row constant that cancels below, and jr is a loop invariant of the lane)
    bool const col0 = j0 == 1 && lane == 0;  // column 0 (jr = -1) enters through lane 0
    i32 run1 = col0 ? h0 - 1 : NEG, run2 = col0 ? h0 - 2 : NEG;
    i32 a1r[CW], a2r[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      i32 const m = max(hmv[c], max(ff[c], oo[c]));
      hh[c] = m;
      a1r[c] = m + jr[c];
      a2r[c] = a1r[c] + jr[c];
      run1 = max(run1, a1r[c]);
      run2 = max(run2, a2r[c]);
    }
    i32 const inc1 = wave_incl_max(run1, NEG), inc2 = wave_incl_max(run2, NEG);
    i32 s1 = wave_shr1(inc1, NEG), s2 = wave_shr1(inc2, NEG);
    if (col0) {
      s1 = h0 - 1;
      s2 = h0 - 2;
    }
    i32 ee[CW], qq[CW];
    // nothing to the left of the first window column when it is not column 1 (lane 0, c == 0 only: from the second
    // column on the running maxima hold the lane's own cells)
    bool const have_left = s1 > -(1 << 28);
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      // (s1 + Q - (j - 1) and max(s2 + G - 2 (j - 1), s1 + Q + G - (j - 2)) of the absolute form)
      i32 q = s1 + (Q_ + 1 - jr[c]);
      i32 e = max(s2 + (G_ + 2 - 2 * jr[c]), q + (G_ + 1));
      if (c == 0) {
        q = have_left ? q : kNegInf;
        e = have_left ? e : kNegInf;
      }
      i32 const m = hh[c];
      s1 = max(s1, a1r[c]);
      s2 = max(s2, a2r[c]);
      ee[c] = e;
      qq[c] = q;
      hh[c] = max(m, max(e, q));
    }
    // ---- (H, E, Q) of the column to the left of this lane's first column ----
    i32 hN = wave_shr1(hh[CW - 1], kNegInf), eN = wave_shr1(ee[CW - 1], kNegInf), qN = wave_shr1(qq[CW - 1], kNegInf);
    if (lane == 0) {
      hN = jb == 1 ? h0 : kNegInf;
      eN = kNegInf;
      qN = kNegInf;
    }
    // ---- decision codes ----
    u32 cd[CW];
    if (fast) {
      i32 hleft = hN, eleft = eN, qleft = qN;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        i32 const ph = H1[c];
        i32 const a1 = F1[c] + E_, a2 = ph + G_, a3 = O1[c] + C_;
        i32 const fv = ff[c], ov = oo[c], hm = hmv[c], h = hh[c], e = ee[c], q = qq[c];
        i32 const b1 = eleft + E_, b2 = hleft + G_, b3 = qleft + C_;
        bool const D = h == hm, U = h == max(fv, ov);
        bool const eu = (h == a1) || ((h != a2) && (h == a3));
        bool const elx = (h == b1) || ((h != b2) && (h == b3));
        bool const lc = (b1 == e) || (b3 == q);
        bool const us = (fv == a1) || (ov == a3);
        u32 code = D ? 0u : (U ? 1u : 2u);
        code |= (!D && (U ? eu : elx)) ? 4u : 0u;
        code |= lc ? 8u : 0u;
        code |= us ? 16u : 32u;
        cd[c] = code;
        hleft = h;
        eleft = e;
        qleft = q;
      }
    } else {
      u32 elmask = 0, lcmask = 0;
      {
        i32 hleft = hN, eleft = eN, qleft = qN;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          i32 const h = hh[c];
          i32 const b1 = eleft + E_, b2 = hleft + G_, b3 = qleft + C_;
          if ((h == b1) || ((h != b2) && (h == b3))) elmask |= 1u << c;
          if ((b1 == ee[c]) || (b3 == qq[c])) lcmask |= 1u << c;
          hleft = h;
          eleft = ee[c];
          qleft = qq[c];
        }
      }
      u32 dmask = 0, umask = 0, eumask = 0, usmask = 0, uhmask = 0;
      u32 xs[CW];
#pragma unroll
      for (int c = 0; c < CW; ++c) xs[c] = 0;
      for (u32 x = 0; x < npe; ++x) {
        u32 const pr = np ? pred_row(g, i, info_cur, x) : 0u;
        i32 th[CW], tf[CW], to[CW], hd;
        fetch(pr, th, tf, to, hd);
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          i32 const a1v = tf[c] + E_, a2v = th[c] + G_, a3v = to[c] + C_, a4v = th[c] + Q_;
          i32 const hv = hd + ((nch == sc[c]) ? M_ : N_);
          hd = th[c];
          i32 const h = hh[c];
          u32 const bit = 1u << c;
          if (!(dmask & bit) && h == hv) {
            dmask |= bit;
            xs[c] |= x;
          }
          bool const t1v = h == a1v, t2v = h == a2v, t3v = h == a3v, t4v = h == a4v;
          if (!(umask & bit) && (t1v || t2v || t3v || t4v)) {
            umask |= bit;
            xs[c] |= x << 2;
            if (t1v || (!t2v && t3v)) eumask |= bit;
          }
          if (np) {
            if (!(usmask & bit) && ((ff[c] == a1v) || (oo[c] == a3v))) {
              usmask |= bit;
              xs[c] |= x << 4;
            }
            if (!(uhmask & bit) && ((ff[c] == a2v) || (oo[c] == a4v))) {
              uhmask |= bit;
              xs[c] |= x << 6;
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        u32 const bit = 1u << c;
        bool const D = dmask & bit, U = umask & bit;
        u32 code = D ? 0u : (U ? 1u : 2u);
        code |= (!D && (U ? ((eumask & bit) != 0) : ((elmask & bit) != 0))) ? 4u : 0u;
        code |= (lcmask & bit) ? 8u : 0u;
        bool const us = usmask & bit, uh = uhmask & bit;
        code |= us ? 16u : (uh ? 32u : 0u);
        code |= (D ? (xs[c] & 3u) : (U ? ((xs[c] >> 2) & 3u) : 0u)) << 6;
        code |= (us ? ((xs[c] >> 4) & 3u) : (uh ? ((xs[c] >> 6) & 3u) : 0u)) << 8;
        cd[c] = code;
      }
    }
    // right exit of this row: the last window column when the haplotype goes on beyond it
    if (lane == 63 && jb + CW - 1 < L) edge = max(edge, hh[CW - 1]);
    {
      u8* const cp = reinterpret_cast<u8*>(codes) + static_cast<size_t>(i) * BW + static_cast<u32>(CW) * lane;
      store_code_bytes<CW>(cp, cd, 0);
      if (!fast) store_code_bytes<CW>(cp + plane, cd, 6);
    }
    if (store) {
      u32 const slot = g.rowslot[i];
      i32* rb = rows + static_cast<size_t>(slot) * 3 * w_stride + static_cast<u32>(CW) * lane;
      if constexpr (CW == 4) {
        *reinterpret_cast<int4*>(rb) = make_int4(hh[0], hh[1], hh[2], hh[3]);
        *reinterpret_cast<int4*>(rb + w_stride) = make_int4(ff[0], ff[1], ff[2], ff[3]);
        *reinterpret_cast<int4*>(rb + 2 * static_cast<size_t>(w_stride)) = make_int4(oo[0], oo[1], oo[2], oo[3]);
      } else if constexpr (CW == 2) {
        *reinterpret_cast<int2*>(rb) = make_int2(hh[0], hh[1]);
        *reinterpret_cast<int2*>(rb + w_stride) = make_int2(ff[0], ff[1]);
        *reinterpret_cast<int2*>(rb + 2 * static_cast<size_t>(w_stride)) = make_int2(oo[0], oo[1]);
      } else {
        rb[0] = hh[0];
        rb[w_stride] = ff[0];
        rb[2 * static_cast<size_t>(w_stride)] = oo[0];
      }
      __threadfence_block();  // other lanes read it back (lane offsets differ between windows)
    }
    if (L >= j0 && L < j0 + BW && static_cast<u32>(lane) == (L - j0) / CW) {
      u32 const cL = (L - j0) % CW;
      i32 v = hh[0];
#pragma unroll
      for (int c = 1; c < CW; ++c) v = (static_cast<u32>(c) == cL) ? hh[c] : v;
      hlast[i] = v;
    } else if (lane == 0 && !(L >= j0 && L < j0 + BW)) {
      hlast[i] = kNegInf;
    }
    // row i takes the place of row i - 2
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      H2[c] = hh[c];
      F2[c] = ff[c];
      O2[c] = oo[c];
    }
  };
  for (u32 i = 1; i <= V; i += 2) {
    row(i, HA, FA, OA, HB, FB, OB);
    if (i + 1 <= V) row(i + 1, HB, FB, OB, HA, FA, OA);
  }
  for (int off = 32; off > 0; off >>= 1) edge = max(edge, __shfl_xor(edge, off));
  if (lane == 0) *edge_out = edge;
}

// ---- the lean banded fill: same cells, same codes as poa_fill_band, a quarter of the instructions ----
// The row loop of poa_fill_band costs ~500 instructions a row whatever its width: half of them scalar -- execution-mask
// bookkeeping for per-lane conditions, uniform branches for rows that might store / slide / hold column 0 / have several
// predecessors -- and a lone wavefront issues one instruction every 4-5 cycles, so the instruction COUNT is the fill's
// latency.  Here every row that is ordinary takes a straight-line path (RI_LEAN, decided once per alignment by
// band_flags in k_msa): a single predecessor = the previous rank (FAST), not read back later (no RI_STORE), not a sink
// (no hlast), a window that does not touch column 0 and sits where the previous row's sits or one lane to its right.
// 97 % of the rows of a haplotype graph.  The other rows take row_gen: poa_fill_band's row without the second
// register row (a predecessor that is not the previous rank comes from the row store: k_msa marks every such row).
//  * no per-lane branches: lane conditions are constant masks (v_cndmask), entering lanes are written with v_writelane;
//  * one scalar load per row (the descriptor of the next row) instead of three;
//  * decision codes from SIGN BITS: a FAST row's code only depends on which argument wins each maximum
//        fA = a1 >= a2, oA = a3 >= a4, FO = fv >= ov, eB = b1 >= b2, qB = b3 >= b4, EQ = e >= q, D = h == hm, U = h == hV
//    (poa_fill's fast path shows that SPOA's ordered equality tests collapse to these; every test is the sign of a
//    difference, h - hm and h - hV being >= 0), so eight differences are shifted into one register with v_alignbit and
//    the 6-bit code is read from a 256-byte table in LDS (one bank per dword: conflict-free): ~19 instead of ~35
//    instructions per cell.  The one cell without a left neighbour (lane 0, first column) gets neighbour values that
//    make eB = qB = false, which is what the equality tests give there (lc = 0); nothing else reads them.
__device__ __forceinline__ u32 lean_code_of(u32 idx) {  // idx bits, 1 = "strictly less": [7] FO [6] fA [5] oA [4] D [3] U [2] EQ [1] eB [0] qB
  bool const FO = !(idx & 128u), fA = !(idx & 64u), oA = !(idx & 32u), D = !(idx & 16u), U = !(idx & 8u), EQ = !(idx & 4u),
             eB = !(idx & 2u), qB = !(idx & 1u);
  u32 code = D ? 0u : (U ? 1u : 2u);
  code |= (!D && (U ? (FO ? fA : oA) : (EQ ? eB : qB))) ? 4u : 0u;
  code |= (eB || qB) ? 8u : 0u;
  code |= (fA || oA) ? 16u : 32u;
  return code;
}
__device__ __forceinline__ u32 sign_into(u32 acc, i32 diff) {  // (acc << 1) | (diff < 0)
  return __builtin_amdgcn_alignbit(acc, static_cast<u32>(diff), 31);
}

template <int CW, class G>
__device__ void poa_fill_lean(G const& g, u32 const w_stride, size_t const plane, u16* codes16, i32* rows, i32* hlast, u32 V,
                              u32 L, int lane, const u8* seq, i32 const edge0, u32 const lut_off, i32* edge_out) {
  static_assert(CW == 1 || CW == 2 || CW == 4, "band tiers: 64, 128 or 256 columns");
  constexpr u32 BW = 64u * CW;
  constexpr i32 NEG = kScanIdent;
  u8* const codes = reinterpret_cast<u8*>(codes16);
  LdsArr<u8> const lut{lut_off};
#pragma unroll
  for (u32 k = 0; k < 4; ++k) lut[4u * lane + k] = static_cast<u8>(lean_code_of(4u * lane + k));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  i32 jr[CW], cq[CW], ce[CW];  // window-relative columns and the per-column constants of Q and E
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    jr[c] = CW * lane + c;
    cq[c] = Q_ + 1 - jr[c];
    ce[c] = G_ + 2 - 2 * jr[c];
  }
  i32 H1[CW], F1[CW], O1[CW];  // the previous row, aligned to the window j0
#pragma unroll
  for (int c = 0; c < CW; ++c) H1[c] = F1[c] = O1[c] = kNegInf;
  u32 sc[CW];
  i32 edge = edge0;
  bool const lane0 = lane == 0, lane63 = lane == 63;
  u32 j0 = g.rowj0[1];
  auto load_sc = [&]() {
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      u32 const j = j0 + static_cast<u32>(CW) * lane + c;
      sc[c] = (j >= 1 && j <= L) ? seq[j - 1] : 0u;
    }
  };
  load_sc();
  // the characters that enter at lane 63 on the next one-lane slide: the words that hold them are fetched one slide ahead
  // (scalar loads, always from inside the haplotype) and taken apart only when they are needed -- extracting them on the
  // spot would put the load's latency on every slide
  u32 scw[CW], scs[CW];
  auto load_sc_in = [&]() {
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      u32 const j = min(j0 + BW + c, L);  // L >= 1
      uintptr_t const a = reinterpret_cast<uintptr_t>(seq + (j - 1));
      // (readfirstlane: folds away on a scalar load and pins the loop-carried word to a scalar register)
      scw[c] = __builtin_amdgcn_readfirstlane(g.uword(reinterpret_cast<const void*>(a & ~uintptr_t(3))));
      scs[c] = __builtin_amdgcn_readfirstlane(static_cast<u32>(a & 3u) * 8u);
    }
  };
  auto sc_in = [&](int c) -> u32 {  // character of column j0 + BW + c (j0: the window BEFORE the slide), 0 beyond the end
    return (j0 + BW + c <= L) ? ((scw[c] >> scs[c]) & 0xFFu) : 0u;
  };
  load_sc_in();
  u8* cp = codes + static_cast<size_t>(BW) + static_cast<u32>(CW) * lane;  // this lane's code bytes of row 1

  // ---- the straight-line row ----
  auto row_lean = [&](u32 const info, auto col0_tag) __attribute__((always_inline)) {
    constexpr bool kCol0 = decltype(col0_tag)::value;  // two straight-line bodies: the ordinary row pays nothing for column 0
    u32 const nch = info & 0xFFu;
    if (!kCol0 && (info & RI_SLIDE)) {
      // lane 0's columns leave the window: exits
      i32 dm = H1[0];
#pragma unroll
      for (int c = 1; c < CW; ++c) dm = max(dm, H1[c]);
      edge = max(edge, lane0 ? dm : kNegInf);
#pragma unroll
      for (int c = 0; c < CW; ++c) {  // lane 63 keeps the fill value: its source lane does not exist
        H1[c] = wave_shl1(H1[c], kNegInf);
        F1[c] = wave_shl1(F1[c], kNegInf);
        O1[c] = wave_shl1(O1[c], kNegInf);
        sc[c] = static_cast<u32>(wave_shl1(static_cast<i32>(sc[c]), static_cast<i32>(sc_in(c))));
      }
      j0 += CW;
      load_sc_in();
    }
    // A row whose window starts at column 1 (info[31:16] = depth + 1, never together with a slide): column 0 is lane 0's left
    // neighbour -- H(i, 0) = h0 and H(i - 1, 0) in closed form of the depth, no E / Q -- and enters the row's prefix maxima
    // with the keys h0 - 1, h0 - 2 (jr = -1).  Every other row: nothing to the left of lane 0 (the fill values).
    u32 const d1 = info >> 16;
    constexpr bool col0 = kCol0;
    i32 const h0 = col0 ? col0_h(d1 - 1u) : kNegInf;
    i32 const hd0 = col0 ? col0_h(d1 - 2u) : kNegInf;  // (the FAST predecessor's depth is one less)
    i32 const s10 = col0 ? h0 - 1 : NEG, s20 = col0 ? h0 - 2 : NEG;
    // vertical + diagonal part
    i32 hd = wave_shr1(H1[CW - 1], hd0);
    i32 ff[CW], oo[CW], hmv[CW], hh[CW], hv[CW], a1r[CW], a2r[CW];
    u32 acc[CW];  // the sign bits of each column, most significant first
    i32 run1 = lane0 ? s10 : NEG, run2 = lane0 ? s20 : NEG;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      i32 const ph = H1[c];
      i32 const a1 = F1[c] + E_, a2 = ph + G_, a3 = O1[c] + C_, a4 = ph + Q_;
      i32 const fv = max(a1, a2), ov = max(a3, a4);
      i32 const hm = hd + ((nch == sc[c]) ? M_ : N_);
      hd = ph;
      ff[c] = fv;
      oo[c] = ov;
      hmv[c] = hm;
      hv[c] = max(fv, ov);
      acc[c] = sign_into(sign_into(static_cast<u32>(fv - ov) >> 31, a1 - a2), a3 - a4);
      i32 const m = max(hm, hv[c]);
      hh[c] = m;
      a1r[c] = m + jr[c];
      a2r[c] = a1r[c] + jr[c];
      run1 = max(run1, a1r[c]);
      run2 = max(run2, a2r[c]);
    }
    // prefix maxima over the window
    i32 const inc1 = wave_incl_max(run1, NEG), inc2 = wave_incl_max(run2, NEG);
    i32 s1 = wave_shr1(inc1, s10), s2 = wave_shr1(inc2, s20);
    i32 ee[CW], qq[CW];
    bool const no_left = lane0 && !col0;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      i32 q = s1 + cq[c];
      i32 e = max(s2 + ce[c], q + (G_ + 1));
      if (c == 0) {  // nothing to the left of the window's first column (unless it is column 0)
        q = no_left ? kNegInf : q;
        e = no_left ? kNegInf : e;
      }
      s1 = max(s1, a1r[c]);
      s2 = max(s2, a2r[c]);
      ee[c] = e;
      qq[c] = q;
      hh[c] = max(hh[c], max(e, q));
    }
    // the column to the left of this lane's first one (lane 0: column 0's H, or the fill value; E / Q values that make
    // eB = qB = false either way, see above)
    i32 hl = wave_shr1(hh[CW - 1], h0), el = wave_shr1(ee[CW - 1], kNegInf - 64), ql = wave_shr1(qq[CW - 1], kNegInf - 64);
    // decision codes: five more sign bits per column, then the table
    u32 code = 0;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      u32 a = acc[c];  // FO fA oA
      i32 const h = hh[c];
      a = sign_into(a, hmv[c] - h);
      a = sign_into(a, hv[c] - h);
      a = sign_into(a, ee[c] - qq[c]);
      a = sign_into(a, (el - hl) + (E_ - G_));
      a = sign_into(a, (ql - hl) + (C_ - Q_));
      code |= static_cast<u32>(lut[a]) << (8 * c);
      hl = h;
      el = ee[c];
      ql = qq[c];
    }
    // right exit of this row: the last window column when the haplotype goes on beyond it
    if (j0 + BW - 1 < L) edge = max(edge, lane63 ? hh[CW - 1] : kNegInf);
    if constexpr (CW == 4) *reinterpret_cast<u32*>(cp) = code;
    else if constexpr (CW == 2) *reinterpret_cast<u16*>(cp) = static_cast<u16>(code);
    else *cp = static_cast<u8>(code);
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      H1[c] = hh[c];
      F1[c] = ff[c];
      O1[c] = oo[c];
    }
  };

  // ---- every other row: poa_fill_band's row with one register row ----
  auto row_gen = [&](u32 const i, u32 const info) __attribute__((always_inline)) {
    u32 const nch = info & 0xFFu, np = (info >> 8) & 7u;
    bool const fast = info & RI_FAST, store = info & RI_STORE;
    i32 const h0 = col0_h(g.rowdepth[i]);
    u32 const j0_new = g.rowj0[i];
    if (j0_new != j0) {
      i32 const sh = (static_cast<i32>(j0_new) - static_cast<i32>(j0)) / CW;  // lanes; > 0: window moves right
      {
        bool const dropped = sh > 0 ? lane < sh : lane >= 64 + sh;
        u32 const jbo = j0 + static_cast<u32>(CW) * lane;
        if (dropped) {
#pragma unroll
          for (int c = 0; c < CW; ++c)
            if (jbo + c >= 1 && jbo + c <= L) edge = max(edge, H1[c]);
        }
      }
      while (j0 < j0_new) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          H1[c] = wave_shl1(H1[c], kNegInf);
          F1[c] = wave_shl1(F1[c], kNegInf);
          O1[c] = wave_shl1(O1[c], kNegInf);
          sc[c] = static_cast<u32>(wave_shl1(static_cast<i32>(sc[c]), static_cast<i32>(sc_in(c))));
        }
        j0 += CW;
        load_sc_in();
      }
      if (j0 > j0_new) {  // rare: the backbone coordinate steps back
        do {
          j0 -= CW;
#pragma unroll
          for (int c = 0; c < CW; ++c) {
            H1[c] = wave_shr1(H1[c], kNegInf);
            F1[c] = wave_shr1(F1[c], kNegInf);
            O1[c] = wave_shr1(O1[c], kNegInf);
            u32 const j = j0 + c;
            sc[c] = static_cast<u32>(wave_shr1(static_cast<i32>(sc[c]), (j >= 1 && j <= L) ? static_cast<i32>(g.ubyte(seq + (j - 1))) : 0));
          }
        } while (j0 > j0_new);
        load_sc_in();
      }
    }
    u32 const jb = j0 + static_cast<u32>(CW) * lane;
    i32 hh[CW], ff[CW], oo[CW], hmv[CW];
    auto left_of = [&](i32 last_col_value, u32 pr) -> i32 {
      i32 const v = wave_shr1(last_col_value, kNegInf);
      if (lane != 0) return v;
      return jb == 1 ? (pr == 0 ? 0 : col0_h(g.rowdepth[pr])) : kNegInf;
    };
    auto fetch_store = [&](u32 pr, i32(&th)[CW], i32(&tf)[CW], i32(&to)[CW], i32& thd) {
      u32 const pj0 = g.rowj0[pr];
      int const d = (static_cast<i32>(j0) - static_cast<i32>(pj0)) / CW;  // this lane reads the stored lane l + d
      int const sl = lane + d;
      bool const in = sl >= 0 && sl < 64;
      u32 const slot = g.rowslot[pr];
      const i32* base = rows + static_cast<size_t>(slot) * 3 * w_stride;
#pragma unroll
      for (int c = 0; c < CW; ++c) th[c] = tf[c] = to[c] = kNegInf;
      if (in) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = base[CW * sl + c];
          tf[c] = base[w_stride + CW * sl + c];
          to[c] = base[2 * static_cast<size_t>(w_stride) + CW * sl + c];
        }
      }
      i32 hd = kNegInf;
      if (jb == 1) {
        hd = col0_h(g.rowdepth[pr]);
      } else if (sl >= 1 && sl <= 64) {
        hd = base[CW * sl - 1];
      }
      thd = hd;
      if (d != 0) {  // stored columns this window does not cover are exits of row pr towards this row
        bool const dropped = d > 0 ? lane < d : lane >= 64 + d;
        if (dropped) {
          u32 const xj = pj0 + static_cast<u32>(CW) * lane;
#pragma unroll
          for (int c = 0; c < CW; ++c)
            if (xj + c <= L) edge = max(edge, base[CW * lane + c]);
        }
      }
    };
    auto fetch = [&](u32 pr, i32(&th)[CW], i32(&tf)[CW], i32(&to)[CW], i32& thd) {
      if (pr == 0) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = row0_h(jb + c);
          tf[c] = kNegInf;
          to[c] = kNegInf;
        }
        thd = row0_h(jb - 1);
      } else if (pr + 1 == i) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          th[c] = H1[c];
          tf[c] = F1[c];
          to[c] = O1[c];
        }
        thd = left_of(H1[CW - 1], pr);
      } else {
        fetch_store(pr, th, tf, to, thd);
      }
    };
    u32 const npe = np ? np : 1u;
    for (u32 x = 0; x < npe; ++x) {
      u32 const pr = np ? pred_row(g, i, info, x) : 0u;
      i32 th[CW], tf[CW], to[CW], hd;
      fetch(pr, th, tf, to, hd);
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        i32 const fv = max(tf[c] + E_, th[c] + G_), ov = max(to[c] + C_, th[c] + Q_);
        i32 const hv = hd + ((nch == sc[c]) ? M_ : N_);
        hd = th[c];
        if (x == 0) {
          ff[c] = fv;
          oo[c] = ov;
          hmv[c] = hv;
        } else {
          ff[c] = max(ff[c], fv);
          oo[c] = max(oo[c], ov);
          hmv[c] = max(hmv[c], hv);
        }
      }
    }
    // row 0 ground that this window does not cover (a node without in-edges hangs off the virtual start row)
    if (np == 0 && j0 + BW <= L) edge = max(edge, row0_h(j0 + BW));
    if (np == 0 && j0 > 1) edge = max(edge, row0_h(1));
    // (column 0 of a row whose window does not start at column 1 is an exit: in edge0, see band_flags)
    bool const col0 = j0 == 1 && lane == 0;  // column 0 (jr = -1) enters through lane 0
    i32 run1 = col0 ? h0 - 1 : NEG, run2 = col0 ? h0 - 2 : NEG;
    i32 a1r[CW], a2r[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      i32 const m = max(hmv[c], max(ff[c], oo[c]));
      hh[c] = m;
      a1r[c] = m + jr[c];
      a2r[c] = a1r[c] + jr[c];
      run1 = max(run1, a1r[c]);
      run2 = max(run2, a2r[c]);
    }
    i32 const inc1 = wave_incl_max(run1, NEG), inc2 = wave_incl_max(run2, NEG);
    i32 s1 = wave_shr1(inc1, NEG), s2 = wave_shr1(inc2, NEG);
    if (col0) {
      s1 = h0 - 1;
      s2 = h0 - 2;
    }
    i32 ee[CW], qq[CW];
    bool const have_left = s1 > -(1 << 28);
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      i32 q = s1 + cq[c];
      i32 e = max(s2 + ce[c], q + (G_ + 1));
      if (c == 0) {
        q = have_left ? q : kNegInf;
        e = have_left ? e : kNegInf;
      }
      i32 const m = hh[c];
      s1 = max(s1, a1r[c]);
      s2 = max(s2, a2r[c]);
      ee[c] = e;
      qq[c] = q;
      hh[c] = max(m, max(e, q));
    }
    i32 hN = wave_shr1(hh[CW - 1], kNegInf), eN = wave_shr1(ee[CW - 1], kNegInf), qN = wave_shr1(qq[CW - 1], kNegInf);
    if (lane == 0) {
      hN = jb == 1 ? h0 : kNegInf;
      eN = kNegInf;
      qN = kNegInf;
    }
    // decision codes: SPOA's tests in SPOA's order (any number of predecessors)
    u32 cd[CW];
    u32 elmask = 0, lcmask = 0;
    {
      i32 hleft = hN, eleft = eN, qleft = qN;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        i32 const h = hh[c];
        i32 const b1 = eleft + E_, b2 = hleft + G_, b3 = qleft + C_;
        if ((h == b1) || ((h != b2) && (h == b3))) elmask |= 1u << c;
        if ((b1 == ee[c]) || (b3 == qq[c])) lcmask |= 1u << c;
        hleft = h;
        eleft = ee[c];
        qleft = qq[c];
      }
    }
    u32 dmask = 0, umask = 0, eumask = 0, usmask = 0, uhmask = 0;
    u32 xs[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) xs[c] = 0;
    for (u32 x = 0; x < npe; ++x) {
      u32 const pr = np ? pred_row(g, i, info, x) : 0u;
      i32 th[CW], tf[CW], to[CW], hd;
      fetch(pr, th, tf, to, hd);
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        i32 const a1v = tf[c] + E_, a2v = th[c] + G_, a3v = to[c] + C_, a4v = th[c] + Q_;
        i32 const hv = hd + ((nch == sc[c]) ? M_ : N_);
        hd = th[c];
        i32 const h = hh[c];
        u32 const bit = 1u << c;
        if (!(dmask & bit) && h == hv) {
          dmask |= bit;
          xs[c] |= x;
        }
        bool const t1v = h == a1v, t2v = h == a2v, t3v = h == a3v, t4v = h == a4v;
        if (!(umask & bit) && (t1v || t2v || t3v || t4v)) {
          umask |= bit;
          xs[c] |= x << 2;
          if (t1v || (!t2v && t3v)) eumask |= bit;
        }
        if (np) {
          if (!(usmask & bit) && ((ff[c] == a1v) || (oo[c] == a3v))) {
            usmask |= bit;
            xs[c] |= x << 4;
          }
          if (!(uhmask & bit) && ((ff[c] == a2v) || (oo[c] == a4v))) {
            uhmask |= bit;
            xs[c] |= x << 6;
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      u32 const bit = 1u << c;
      bool const D = dmask & bit, U = umask & bit;
      u32 code = D ? 0u : (U ? 1u : 2u);
      code |= (!D && (U ? ((eumask & bit) != 0) : ((elmask & bit) != 0))) ? 4u : 0u;
      code |= (lcmask & bit) ? 8u : 0u;
      bool const us = usmask & bit, uh = uhmask & bit;
      code |= us ? 16u : (uh ? 32u : 0u);
      code |= (D ? (xs[c] & 3u) : (U ? ((xs[c] >> 2) & 3u) : 0u)) << 6;
      code |= (us ? ((xs[c] >> 4) & 3u) : (uh ? ((xs[c] >> 6) & 3u) : 0u)) << 8;
      cd[c] = code;
    }
    if (lane == 63 && jb + CW - 1 < L) edge = max(edge, hh[CW - 1]);
    store_code_bytes<CW>(cp, cd, 0);
    if (!fast) store_code_bytes<CW>(cp + plane, cd, 6);
    if (store) {
      u32 const slot = g.rowslot[i];
      i32* rb = rows + static_cast<size_t>(slot) * 3 * w_stride + static_cast<u32>(CW) * lane;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        rb[c] = hh[c];
        rb[w_stride + c] = ff[c];
        rb[2 * static_cast<size_t>(w_stride) + c] = oo[c];
      }
      __threadfence_block();  // other lanes read it back (lane offsets differ between windows)
    }
    if (L >= j0 && L < j0 + BW && static_cast<u32>(lane) == (L - j0) / CW) {
      u32 const cL = (L - j0) % CW;
      i32 v = hh[0];
#pragma unroll
      for (int c = 1; c < CW; ++c) v = (static_cast<u32>(c) == cL) ? hh[c] : v;
      hlast[i] = v;
    } else if (lane == 0 && !(L >= j0 && L < j0 + BW)) {
      hlast[i] = kNegInf;
    }
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      H1[c] = hh[c];
      F1[c] = ff[c];
      O1[c] = oo[c];
    }
    // Leave no vector load of this path in flight: where the two paths meet the compiler cannot tell which registers a load
    // may still be writing and would wait for vmcnt(0) at the head of EVERY row -- that is, for the previous row's code
    // store, a store round trip per row.
    __builtin_amdgcn_s_waitcnt(0);
  };

  u32 info = g.rowinfo[1];
  __builtin_amdgcn_s_waitcnt(0);  // (the same for the loads above: nothing in flight when the loop is entered)
#ifdef MA_PROFILE
  unsigned long long t_lean = 0, t_gen = 0, n_gen = 0;
  unsigned long long const t_fill0 = __builtin_amdgcn_s_memtime();
#endif
  for (u32 i = 1; i <= V; ++i) {
    u32 const cur = info;
    if (i < V) info = g.rowinfo[i + 1];  // the next row's descriptor: a scalar load, a row ahead
#ifdef MA_PROFILE
    unsigned long long const q0 = __builtin_amdgcn_s_memtime();
#endif
    if (__builtin_expect((cur & RI_LEAN) != 0, 1)) {
      if (__builtin_expect((cur >> 16) == 0, 1)) row_lean(cur, std::false_type{});
      else row_lean(cur, std::true_type{});
    } else {
      row_gen(i, cur);
    }
#ifdef MA_PROFILE
    unsigned long long const q1 = __builtin_amdgcn_s_memtime();
    if (cur & RI_LEAN) t_lean += q1 - q0; else { t_gen += q1 - q0; n_gen++; }
#endif
    cp += BW;
  }
#ifdef MA_PROFILE
  if (lane == 0) {  // (s_memtime ticks: 100 MHz) [8] lean rows' time, [9] other rows' time, [10] other rows, [11] fills, [12] rows, [13] whole fills
    atomicAdd(&g_prof[8], t_lean);
    atomicAdd(&g_prof[9], t_gen);
    atomicAdd(&g_prof[10], n_gen);
    atomicAdd(&g_prof[11], 1ull);
    atomicAdd(&g_prof[12], static_cast<unsigned long long>(V));
    atomicAdd(&g_prof[13], __builtin_amdgcn_s_memtime() - t_fill0);
  }
#endif
  for (int off = 32; off > 0; off >>= 1) edge = max(edge, __shfl_xor(edge, off));
  if (lane == 0) *edge_out = edge;
}

// ---- traceback (wave 0, all lanes carry the same state): SisdAlignmentEngine::Convex backtrack ----
struct EdgeVals {
  i32 h, f, e, o, q;
};
__device__ __forceinline__ EdgeVals edge_vals(GL const& g, u32 i, u32 j) {  // cells of row 0 / column 0
  EdgeVals v;
  if (i == 0 && j == 0) {
    v.h = v.f = v.e = v.o = v.q = 0;
  } else if (i == 0) {
    v.q = Q_ + static_cast<i32>(j - 1) * C_;
    v.e = G_ + static_cast<i32>(j - 1) * E_;
    v.h = max(v.q, v.e);
    v.f = v.o = kNegInf;
  } else {
    u32 const d = g.rowdepth[i];
    v.o = col0_o(d);
    v.f = col0_f(d);
    v.h = max(v.o, v.f);
    v.e = v.q = kNegInf;
  }
  return v;
}

// bw: columns of the band the codes were written for (64, 128 or 256); 0 = the full row-synchronous fill
__device__ __forceinline__ u32 poa_traceback(GL const& g, const u16* codes16, size_t const plane, u32 cw, u32 V, u32 L, u32 best_row,
                             bool have_end, int lane, u32 bw) {
  const u8* const codes = reinterpret_cast<const u8*>(codes16);
  bool const band = bw != 0;
  u32 const nl = (L + cw - 1) / cw;
  bool off_band = false;  // band mode: a cell outside its row's window was needed -> the caller falls back
  auto code_at = [&](u32 i, u32 j) -> u32 {  // i >= 1, j >= 1
    if (band) {
      u32 const j0 = g.rowj0[i];
      if (j < j0 || j >= j0 + bw) {
        off_band = true;
        return 2u;  // a plain "left" move: lets the caller's loops terminate
      }
      size_t const at = static_cast<size_t>(i) * bw + (j - j0);
      u32 const a = codes[at];
      return (g.rowinfo[i] & RI_FAST) ? a : (a | (static_cast<u32>(codes[plane + at]) << 6));
    }
    size_t const at = static_cast<size_t>(i) * (nl * cw) + (j - 1);
    u32 const a = codes[at];
    return (g.rowinfo[i] & RI_FAST) ? a : (a | (static_cast<u32>(codes[plane + at]) << 6));
  };
  u32 naln = 0;
  bool overflow = false;
  auto push1 = [&](u32 node1, u32 pos1) {  // node id + 1 | 0, seq pos + 1 | 0
    if (naln + 1 >= g.aln_cap) {
      overflow = true;
      return;
    }
    if (lane == 0) {
      g.aln[2 * naln] = static_cast<u16>(node1);
      g.aln[2 * naln + 1] = static_cast<u16>(pos1);
    }
    naln++;
  };
  u32 i = have_end ? best_row : 0u, j = have_end ? L : 0u;
  u32 prev_i = 0, prev_j = 0;
  while (!(i == 0 && j == 0) && !overflow && !__ballot(off_band)) {
    if (i >= 1 && j >= 1) {
      // 64 cells down the diagonal at once: a run of diagonal moves over rank-consecutive rows
      {
        u32 const k = static_cast<u32>(lane);
        bool ok = false;
        if (i >= k + 2 && j >= k + 1) ok = ((code_at(i - k, j - k) & 3u) == 0u) && (g.rowinfo[i - k] & RI_FAST);
        unsigned long long const m = __ballot(ok);
        u32 const run = m == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~m));
        if (run > 0) {
          if (naln + run + 1 >= g.aln_cap) {
            overflow = true;
            break;
          }
          if (k < run) {
            g.aln[2 * (naln + k)] = static_cast<u16>(g.rank2node[i - k - 1] + 1u);
            g.aln[2 * (naln + k) + 1] = static_cast<u16>(j - k);
          }
          naln += run;
          i -= run;
          j -= run;
          prev_i = i;
          prev_j = j;
          continue;
        }
      }
      u32 const cd = code_at(i, j);
      u32 const info = g.rowinfo[i];
      u32 const np = (info >> 8) & 7u;
      u32 const node = g.rank2node[i - 1];
      u32 const kind = cd & 3u;
      bool const ext = cd & 4u;
      u32 const x = (cd >> 6) & 3u;
      if (kind == 0) {
        prev_i = np ? pred_row(g, i, info, x) : 0u;
        prev_j = j - 1;
        push1(node + 1, j);
        i = prev_i;
        j = prev_j;
      } else if (kind == 1) {
        prev_i = np ? pred_row(g, i, info, x) : 0u;
        prev_j = j;
        push1(node + 1, 0);
        i = prev_i;
        if (ext) {  // walk up the F/O extension run in column j
          while (!overflow && i >= 1) {
            {
              u32 const k = static_cast<u32>(lane);
              bool ok = false;
              if (i >= k + 2) ok = (code_at(i - k, j) & 16u) && (g.rowinfo[i - k] & RI_FAST);
              unsigned long long const m = __ballot(ok);
              u32 const run = m == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~m));
              if (run > 0) {
                if (naln + run + 1 >= g.aln_cap) {
                  overflow = true;
                  break;
                }
                if (k < run) {
                  g.aln[2 * (naln + k)] = static_cast<u16>(g.rank2node[i - k - 1] + 1u);
                  g.aln[2 * (naln + k) + 1] = 0;
                }
                naln += run;
                i -= run;
                prev_i = i;
                continue;  // i >= 2 - 1 >= 1 here: the last consumed row was >= 2
              }
            }
            u32 const c2 = code_at(i, j);
            u32 const info2 = g.rowinfo[i];
            bool const us = c2 & 16u, uh = c2 & 32u;
            u32 const pi = (us || uh) ? pred_row(g, i, info2, (c2 >> 8) & 3u) : 0u;
            push1(g.rank2node[i - 1] + 1u, 0);
            prev_i = pi;
            i = pi;
            if (!us || i == 0) break;
          }
        }
      } else {
        prev_i = i;
        prev_j = j - 1;
        push1(0, j);
        j = prev_j;
        if (ext) {  // left extension run along row i: entries (none, pos) while the E/Q chain continues
          while (!overflow && j >= 1) {
            u32 const k = static_cast<u32>(lane);
            bool cont = false;
            if (j >= k + 1) cont = code_at(i, j - k) & 8u;
            unsigned long long const m = __ballot(cont);
            u32 const nc = m == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~m));
            u32 const cnt = nc < 64u ? nc + 1u : 64u;
            if (naln + cnt + 1 >= g.aln_cap) {
              overflow = true;
              break;
            }
            if (k < cnt) {
              g.aln[2 * (naln + k)] = 0;
              g.aln[2 * (naln + k) + 1] = static_cast<u16>(j - k);  // pos + 1 with pos = j - k - 1
            }
            naln += cnt;
            j -= cnt;
            if (nc < 64u) break;
          }
        }
      }
      continue;
    }
    // ---- row 0 / column 0: closed-form values, same tests in the same order ----
    i32 const Hij = edge_vals(g, i, j).h;
    bool found = false, ext_left = false, ext_up = false;
    u32 const node = i ? g.rank2node[i - 1] : 0u;
    u32 const info = i ? g.rowinfo[i] : 0u;
    u32 const np = i ? ((info >> 8) & 7u) : 0u;
    if (i != 0) {
      for (u32 x = 0; x < (np ? np : 1u); ++x) {
        u32 const pi = np ? pred_row(g, i, info, x) : 0u;
        EdgeVals const pv = edge_vals(g, pi, j);
        bool ok = (ext_up |= (Hij == pv.f + E_));
        if (!ok) ok = Hij == pv.h + G_;
        if (!ok) ok = (ext_up |= (Hij == pv.o + C_));
        if (!ok) ok = Hij == pv.h + Q_;
        if (ok) {
          prev_i = pi;
          prev_j = j;
          found = true;
          break;
        }
      }
    }
    if (!found && j != 0) {
      EdgeVals const lv = edge_vals(g, i, j - 1);
      bool ok = (ext_left |= (Hij == lv.e + E_));
      if (!ok) ok = Hij == lv.h + G_;
      if (!ok) ok = (ext_left |= (Hij == lv.q + C_));
      if (!ok) ok = Hij == lv.h + Q_;
      if (ok) {
        prev_i = i;
        prev_j = j - 1;
        found = true;
      }
    }
    push1(i != prev_i ? node + 1 : 0u, j != prev_j ? j : 0u);
    i = prev_i;
    j = prev_j;
    if (ext_left) {
      while (!overflow) {
        push1(0, j);
        --j;
        bool const e_stop = edge_vals(g, i, j).e + E_ != edge_vals(g, i, j + 1).e;
        bool const q_stop = edge_vals(g, i, j).q + C_ != edge_vals(g, i, j + 1).q;
        if ((e_stop && q_stop) || j == 0) break;
      }
    } else if (ext_up) {
      while (!overflow && i >= 1) {
        bool stop = true;
        prev_i = 0;
        u32 const nd2 = g.rank2node[i - 1];
        u32 const info2 = g.rowinfo[i];
        u32 const np2 = (info2 >> 8) & 7u;
        EdgeVals const cv = edge_vals(g, i, j);
        for (u32 x = 0; x < np2; ++x) {
          u32 const pr = pred_row(g, i, info2, x);
          EdgeVals const pv = edge_vals(g, pr, j);
          if (cv.f == pv.f + E_ || cv.o == pv.o + C_) {
            prev_i = pr;
            stop = false;
            break;
          }
        }
        if (stop) {
          for (u32 x = 0; x < np2; ++x) {
            u32 const pr = pred_row(g, i, info2, x);
            i32 const hp2 = edge_vals(g, pr, j).h;
            if (cv.f == hp2 + G_ || cv.o == hp2 + Q_) {
              prev_i = pr;
              break;
            }
          }
        }
        push1(nd2 + 1u, 0);
        i = prev_i;
        if (stop || i == 0) break;
      }
    }
  }
  if (__ballot(off_band)) {
    if (lane == 0) ST.band_fail = 1;
    return 0;
  }
  if (overflow && lane == 0) ST.overflow = 1;
  return naln;
}

// band window of a row: 64 cwb columns centred on the node's backbone coordinate, j0 = 1 mod cwb
__device__ __forceinline__ u16 band_j0(u32 npos, u32 L, u32 cwb) {
  u32 const bw = 64u * cwb;
  i32 const want = static_cast<i32>(npos) + 1 - static_cast<i32>(bw / 2);
  u32 const j0 = ((static_cast<u32>(max(want, 1)) - 1u) & ~(cwb - 1u)) + 1u;
  u32 const jmax = L > bw ? (((L - bw) + cwb - 1u) & ~(cwb - 1u)) + 1u : 1u;
  return static_cast<u16>(min(j0, jmax));
}

// Per-row flags of a band tier (after rowj0, RI_STORE and rowdepth are final): which rows take poa_fill_lean's
// straight-line path, and the exits that are known before the fill (column 0 of a row is an exit when the row's window
// does not start at column 1; its value is closed-form).
__device__ __forceinline__ void band_flags(GL const& g, u32 V, u32 cwb, int tid) {
  int const lane = tid & 63, wave = tid >> 6;
  i32 e0 = kNegInf;
  for (u32 i = 1 + tid; i <= V; i += kT) {
    u32 info = g.rowinfo[i] & ~(RI_LEAN | RI_SLIDE);
    if (info & RI_FAST) info &= 0xFFFFu;  // (a retried tier: the depth a column-0 lean row carried in its upper half)
    u32 const j0 = g.rowj0[i];
    if (j0 > 1) e0 = max(e0, col0_h(g.rowdepth[i]));
    if (i >= 2 && cwb) {
      u32 const jp = g.rowj0[i - 1];
      if ((info & RI_FAST) && !(info & RI_STORE) && g.nout[g.rank2node[i - 1]] != 0) {
        if (j0 > 1 && (j0 == jp || j0 == jp + cwb)) {
          info = (info & 0xFFFFu) | RI_LEAN;
          if (j0 != jp) info |= RI_SLIDE;
        } else if (j0 == 1 && jp == 1) {
          // (round 6) the window starts at column 1, like the previous row's: column 0 -- closed form of the row's depth --
          // is lane 0's left neighbour.  Bits [31:16] (free in a FAST row) carry depth + 1; the ~64 first rows of every
          // alignment took the any-shape row body for this alone (4200 cycles a row against 850)
          info = (info & 0xFFFFu) | RI_LEAN | ((static_cast<u32>(g.rowdepth[i]) + 1u) << 16);
        }
      }
    }
    g.rowinfo[i] = info;
  }
  for (int off = 32; off > 0; off >>= 1) e0 = max(e0, __shfl_xor(e0, off));
  if (lane == 0) ST.red_v[wave] = e0;
  __syncthreads();
  if (tid == 0) ST.edge0 = max(max(ST.red_v[0], ST.red_v[1]), max(ST.red_v[2], ST.red_v[3]));
  __syncthreads();
}

struct MsaArgs {
  DBatch b;
  ma_asm_out_t a;
  ma_var_out_t o;
  PoaWs ws;
  ma_params_t prm;
  int win0;
  u32 round;   // split mode: 0 = start, > 0 = resume from the LDS image
  u32 finish;  // split mode: last launch -- every fill still to do runs inside k_msa (no more yields)
};

// A batch with a component of more than 16 haplotypes runs in two passes (launch_msa): the common kernels over the windows
// that fit 16-bit label masks, the LAB32 kernels -- 8 more bytes of LDS per node: one workgroup per CU -- over the few that do not.
__device__ __forceinline__ bool poa_window_skipped(MsaArgs const& A, int w) {
  if ((A.a.win_status[w] & MA_W_NO_HAPLOTYPE) || A.a.win_ncomp[w] == 0) return true;
  if (A.ws.pass == 0) return false;
  bool wide = false;
  u32 const nc = A.a.win_ncomp[w];
  for (u32 c = 0; c < nc; ++c) wide = wide || A.a.comp_nhaps[static_cast<size_t>(w) * A.prm.max_comps + c] > 16;
  return wide != (A.ws.pass == 2);
}

// One window's POA up to its next banded fill (returns kMsaYield: the LDS block is saved, the pending alignment's descriptors
// are in the image) or to its end (kMsaDone).  `resume`: continue from the image after a fill.  Runs as the body of k_msa
// (host-counted rounds, MA_POA_SCHED=0) and as the G job of the persistent kernel k_poa.
constexpr u32 kMsaDone = 0, kMsaYield = 1;
template <int CWMAX, bool LAB32>
__device__ __forceinline__ u32 msa_window(MsaArgs const& A, int const lw, bool const resume_in) {
  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int const w = A.win0 + lw;
  ma_params_t const& P = A.prm;
  PoaWs const& ws = A.ws;
  int const MC = P.max_comps, MH = P.max_haps, ML = P.max_hap_len, MV = P.max_vars, MA = P.max_alts,
            MP = P.max_allele_bytes;

  u32 const ncomp = (A.a.win_status[w] & MA_W_NO_HAPLOTYPE) ? 0u : A.a.win_ncomp[w];
  if (ncomp == 0) {
    if (tid == 0) A.o.win_nvars[w] = 0;
    return kMsaDone;
  }
  if (poa_window_skipped(A, w)) return kMsaDone;  // the other pass's window
  u32 const PN = ws.pn;
  GL const g = poa_carve(PN, ws.max_l, LAB32 ? 1u : 0u);
  u16* const codes = reinterpret_cast<u16*>(reinterpret_cast<u8*>(ws.codes) + static_cast<size_t>(lw) * ws.code_cells * 2);
  i32* const rows = ws.rows + static_cast<size_t>(lw) * ws.row_cells;
  size_t const fslot = ws.full_by_worker ? static_cast<size_t>(blockIdx.x) : static_cast<size_t>(lw);
  u16* const fcodes = reinterpret_cast<u16*>(reinterpret_cast<u8*>(ws.fcodes) + fslot * ws.fcode_cells * 2);
  i32* const frows = ws.frows + fslot * ws.frow_cells;
  i32* const hlast = ws.hlast + static_cast<size_t>(lw) * (PN + 8);
  // raw-allele scratch of the bubble walk: the alignment scratch in LDS (free by then) while a bubble's alleles fit,
  // the code area in HBM beyond that (every byte thread 0 reads back from HBM is a round trip of its serial walk)
  u8* const raw_hbm = reinterpret_cast<u8*>(codes);
  u8* const raw_lds = &ma_lds[g.rowinfo.off];
  u32 const raw_lds_room = (static_cast<u32>(poa_lds_bytes(PN, ws.max_l, LAB32 ? 1u : 0u)) - 16u - g.rowinfo.off) / g.max_seq();  // bytes per haplotype
  u32 const raw_lds_cap = ws.raw_cap ? min(ws.raw_cap, raw_lds_room) : raw_lds_room;

  // Split mode: the banded fill of an alignment runs in k_msa_band (one wavefront per window, a dozen windows per
  // CU instead of two) between two launches of this kernel.  This kernel then works like a coroutine: when an
  // alignment is ready for the band it saves its whole LDS block (graph + state) to HBM and returns; the next
  // launch restores the block and carries on right after the fill.
  u32* const img = ws.split ? reinterpret_cast<u32*>(ws.img) + static_cast<size_t>(lw) * ws.img_words : nullptr;
  bool resume = false;
#ifdef MA_PROFILE
  unsigned long long const t_in = __builtin_amdgcn_s_memtime();
#endif
  if (resume_in) {
    if (reinterpret_cast<const WgState*>(img)->done) return kMsaDone;
    {  // 16 bytes per thread and load, four loads in flight (img_words is a multiple of 64, the image 256-byte aligned)
      const uint4* src = reinterpret_cast<const uint4*>(img);
      uint4* dst = reinterpret_cast<uint4*>(ma_lds);
      // (the block's last array, the alignment path buffer `aln`, is scratch before the fill -- the list of slow rows -- and
      //  written afresh by the traceback after it: neither saved nor restored, 9 of the block's 80 KB)
      u32 const nq = min(ws.img_words / 4u, (g.aln.off + 15u) / 16u);
      for (u32 i0 = tid; i0 < nq; i0 += 4u * kT) {
        uint4 v[4];
#pragma unroll
        for (u32 u = 0; u < 4; ++u) v[u] = i0 + u * kT < nq ? src[i0 + u * kT] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (u32 u = 0; u < 4; ++u)
          if (i0 + u * kT < nq) dst[i0 + u * kT] = v[u];
      }
    }
    __syncthreads();
    resume = true;
  }
  // thread 0's running output state
  u32 nvars = 0, pool = 0;
  bool overflow = false;
  if (resume) {
    if (tid == 0) {
      nvars = ST.nvars;
      pool = ST.pool;
      overflow = ST.var_overflow != 0;
    }
  } else if (tid == 0) {
    ST.win_overflow = 0;
    ST.done = 0;
    ST.filled = 0;
    ST.pending = 0;
    ST.c_cur = ST.h_cur = 0;
  }
  __syncthreads();
  PROF_T0();
#ifdef MA_PROFILE
  if (tid == 0) atomicAdd(&g_prof[7], _t0 - t_in);  // image restore
#endif
  bool yielded = false;
  u32 const c_start = resume ? ST.c_cur : 0u;

  for (u32 c = c_start; c < ncomp; ++c) {
    bool const rc = resume && c == c_start;  // re-entering the component the pending alignment belongs to
    __syncthreads();
    if (ST.win_overflow) break;
    size_t const ci = static_cast<size_t>(w) * MC + c;
    u32 const hap0 = A.a.comp_hap0[ci], nh = A.a.comp_nhaps[ci];
    if (tid == 0 && !rc) {
      ST.nn = ST.nseq = ST.nrank = 0;
      ST.overflow = nh > g.max_seq() ? 1u : 0u;  // label masks are 16 / 32 bit (launch_msa sends wider components to the LAB32 pass)
    }
    u32 const h_start = rc ? ST.h_cur : 0u;
    for (u32 h = h_start; h < nh; ++h) {
      bool const rh = rc && h == h_start;  // the alignment k_msa_band has just filled
      size_t const hi = static_cast<size_t>(w) * MH + hap0 + h;
      const u8* seq = A.a.hap_bases + hi * ML;
      u32 const L = A.a.hap_len[hi];
      __syncthreads();
      if (tid == 0 && rh) ST.pending = 0;
      if (tid == 0 && !rh) {
        u32 mode = 0;
        if (!ST.overflow && L > 0) {
          if (ST.nn == 0) {
            if (L > PN) ST.overflow = 1; else mode = 1;
          } else {
            u32 const V = ST.nrank;
            u32 const cw = L <= 1024 ? 4u : (L <= 1536 ? 6u : (L <= 2048 ? 8u : (L <= 3072 ? 12u : 16u)));
            u32 const nl = (L + cw - 1) / cw;
            if (L > 4096 || V > PN || static_cast<size_t>(V + 1) * nl * cw > ws.fcode_cells) {
              ST.overflow = 1;
            } else {
              mode = 2;
              ST.V = V;
              ST.cw = cw;
              // first band tier (columns per lane: 64 / 128 / 256 columns), 0 = full fill only
              // a haplotype that is much longer or shorter than the component's backbone (its first sequence, the REF
              // anchor) carries an indel of at least that size: the 128-column band cannot hold its path, so it starts at the
              // 256-column tier instead of failing the narrow certificate first (a round, i.e. a fill's latency, saved;
              // the tiers are exact, the result is the same)
              u32 const Lref = A.a.hap_len[static_cast<size_t>(w) * MH + hap0];
              u32 const dl = L > Lref ? L - Lref : Lref - L;
              u32 const t0 = (ws.tier0 < 4u && dl > 48u && !ws.no_wide_start) ? 4u : ws.tier0;
              ST.band = (ws.use_band && L >= 400 && static_cast<size_t>(V + 1) * 256 <= ws.code_cells) ? t0 : 0u;
            }
          }
        }
        ST.mode = mode;
        ST.L = L;
        ST.nslow = 0;
        ST.naln = 0;
      }
      __syncthreads();
      u32 const mode = ST.mode;
      if (mode == 0) continue;
      if (mode == 1) {
        // first sequence of the component: a linear chain whose topological order is the identity
        u32 const lab = 1u << ST.nseq;
        for (u32 i = tid; i < L; i += kT) {
          g.nchar[i] = seq[i];
          g.nin[i] = i > 0 ? 1 : 0;
          g.nout[i] = i + 1 < L ? 1 : 0;
          g.nal[i] = 0;
          g.in_tail[i * kPE] = static_cast<u16>(i - 1);
          g.out_head[i * kPE] = static_cast<u16>(i + 1);
          g.lab_set(i * kPE, lab);
          g.rank2node[i] = static_cast<u16>(i);
          g.node2rank[i] = static_cast<u16>(i);
          g.npos[i] = static_cast<u16>(i);
        }
        __syncthreads();
        if (tid == 0) {
          ST.seq_first[ST.nseq] = 0;
          ST.nseq++;
          ST.nn = L;
          ST.nrank = L;
        }
        PROF_ACC(6);
        continue;
      }
      // ---- mode 2: align the haplotype to the graph ----
      u32 const V = ST.V, cw = ST.cw;
      // ---- alignments that need no fill: the graph is still the component's first sequence R (a linear chain in rank
      //      order, V nodes), the haplotype Q (L bases) is R with at most two substitutions, or with ONE indel and nothing else.
      // (a) L == V, <= 2 substitutions.  For every prefix pair (i, i) the diagonal scores -6 per substitution, >= -12, and
      //     any other path to (i, i) has a vertical and a horizontal gap, <= -12: H(i, i) is the diagonal's score, so every
      //     diagonal cell passes the backtrack's FIRST test (H == H(diagonal predecessor) + score) and SPOA retraces exactly
      //     the diagonal from the only end cell, ties or not.  (Three substitutions could lose to a mismatch-free path with
      //     two unit gaps: those take the DP.)
      // (b) |V - L| = l > 0 and common prefix + common suffix >= min(V, L): Q is R with one insertion / deletion of l bases
      //     at any place a in [min - lcs, lcp].  g(l) = max(G + (l-1) E, Q + (l-1) C) is the best score of ANY cell pair
      //     whose lengths differ by l (gap costs are superadditive, substitutions cost), and it is reached by every prefix
      //     pair on the shifted diagonal that has a valid place a' <= its length: from the end cell the first test passes
      //     (match, H equal) down to the LEFTMOST place a_min = max(0, min - lcs); there it fails (a base mismatch, by
      //     minimality, and no neighbour scores above g(l)), the cell's only optimal entry is the gap (a path entering it
      //     the other way needs l + 1 and 1 gap bases), the gap run's extension flags hold for exactly its l cells (opening
      //     anywhere but after the exact prefix, H = 0, costs a second gap), and the exact prefix is retraced diagonally.
      //     So SPOA's backtrack yields: shifted diagonal, the l gap cells at a_min, main diagonal -- written down here.
      // About half of all alignments on the bench workload.  MA_POA_NO_DIRECT sends them through the fill (tests).
      bool direct = false;
      if (!rh && ST.nseq == 1 && !ws.no_direct && L > 0 && V > 0) {
        u32 const m = min(L, V);
        if (L == V) {
          u32 mism = 0;
          for (u32 i = tid; i < L; i += kT) mism += g.nchar[g.rank2node[i]] != seq[i];
          u32 total = 0;
          (void)block_excl_scan(mism, tid, total);
          direct = total <= 2;
          if (direct) {
            for (u32 k = tid; k < L; k += kT) {  // the path as the traceback stores it: end first, (node + 1, column)
              g.aln[2 * k] = static_cast<u16>(g.rank2node[V - 1 - k] + 1u);
              g.aln[2 * k + 1] = static_cast<u16>(L - k);
            }
            if (tid == 0) ST.naln = L;
          }
        } else {
          u32 p1 = m, s1 = m;  // first mismatch from the front / from the back
          for (u32 i = tid; i < m; i += kT) {
            if (g.nchar[g.rank2node[i]] != seq[i]) p1 = min(p1, i);
            if (g.nchar[g.rank2node[V - 1 - i]] != seq[L - 1 - i]) s1 = min(s1, i);
          }
          u32 const lcp = block_min(p1, tid), lcs = block_min(s1, tid);
          direct = lcp + lcs >= m;
          if (direct) {
            u32 const gl = L > V ? L - V : V - L;
            u32 const a0 = m > lcs ? m - lcs : 0u;  // leftmost place of the indel
            u32 const nd = m - a0;                    // cells on the shifted diagonal
            bool const del = V > L;                   // graph nodes without a base: vertical gap
            u32 const total = nd + gl + a0;
            for (u32 k = tid; k < total; k += kT) {
              u32 node1, col;
              if (k < nd) {
                node1 = g.rank2node[V - 1 - k] + 1u;
                col = L - k;
              } else if (k < nd + gl) {
                u32 const t = k - nd;  // t-th gap cell from the end
                node1 = del ? g.rank2node[a0 + gl - 1 - t] + 1u : 0u;
                col = del ? 0u : a0 + gl - t;
              } else {
                u32 const t = k - nd - gl;
                node1 = g.rank2node[a0 - 1 - t] + 1u;
                col = a0 - t;
              }
              g.aln[2 * k] = static_cast<u16>(node1);
              g.aln[2 * k + 1] = static_cast<u16>(col);
            }
            if (tid == 0) ST.naln = total;
          }
        }
      }
      if (ws.dstats && tid == 0 && !rh) {
        atomicAdd(&ws.dstats[10], 1ull);
        if (direct) atomicAdd(&ws.dstats[11], 1ull);
      }
      if (!direct) {
      if (!rh) {
      // per-row descriptors; which rows must be kept in HBM
      for (u32 i = tid; i <= V + 1; i += kT) g.rowslot[i] = 0;
      __syncthreads();
      for (u32 i = 1 + tid; i <= V; i += kT) {
        u32 const node = g.rank2node[i - 1];
        u32 const np = g.nin[node];
        u32 pr[kPE];
        for (u32 x = 0; x < kPE; ++x) pr[x] = x < np ? static_cast<u32>(g.node2rank[g.in_tail[node * kPE + x]]) + 1u : 0u;
        u32 info = static_cast<u32>(g.nchar[node]) | (np << 8);
        if (np == 1 && pr[0] + 1 == i) {
          info |= RI_FAST;
        } else if (np > 0) {
          u32 const idx = atomicAdd(&ST.nslow, 1u);
          if (idx < kSlowCap) {
            for (u32 x = 0; x < np; ++x) g.slowpred[idx * kPE + x] = static_cast<u16>(pr[x]);
            info |= RI_SLOWTAB | (idx << 16);
          }
        }
        for (u32 x = 0; x < np; ++x)
          if (pr[x] + 1 != i) g.rowslot[pr[x]] = 1;  // (poa_fill / poa_fill_band still take rank - 2 from registers)
        g.rowinfo[i] = info;
        g.rowj0[i] = band_j0(g.npos[node], L, ST.band ? ST.band : 4u);
      }
      __syncthreads();
      {
        u32 const per = (V + kT) / kT;  // rows 1 .. V in contiguous chunks
        u32 const lo = min(1 + tid * per, V + 1), hi2 = min(lo + per, V + 1);
        u32 cnt = 0, nslow = 0, last_slow = 0;
        for (u32 i = lo; i < hi2; ++i) {
          cnt += g.rowslot[i];
          if (!(g.rowinfo[i] & RI_FAST)) {
            nslow++;
            last_slow = i;
          }
        }
        u32 total = 0, total_slow = 0;
        u32 slot = block_excl_scan(cnt, tid, total);
        u32 at = block_excl_scan(nslow, tid, total_slow);
        u32 head = block_excl_scan_max(last_slow, tid);  // last row before this chunk that is not FAST
        for (u32 i = lo; i < hi2; ++i) {
          if (g.rowslot[i]) {
            g.rowslot[i] = static_cast<u16>(slot++);
            g.rowinfo[i] |= RI_STORE;
          } else {
            g.rowslot[i] = 0xFFFFu;
          }
          if (!(g.rowinfo[i] & RI_FAST)) {
            head = i;
            g.aln[at++] = static_cast<u16>(i);  // the alignment path buffer is free until the traceback
          }
          g.runhead[i] = static_cast<u16>(head);
        }
        if (total > ws.row_slots && tid == 0) ST.overflow = 1;
        __syncthreads();
        // column 0 of the DP only depends on every node's minimum hop distance d from a root
        // (SisdAlignmentEngine::Initialize: O0 = Q + C d, F0 = G + E d).  Rows that are not FAST take it
        // from their predecessors, serially in rank order; FAST rows just count up from their run head.
        if (tid == 0) {
          for (u32 k = 0; k < total_slow; ++k) {
            u32 const r = g.aln[k];
            u32 const inf = g.rowinfo[r];
            u32 const np = (inf >> 8) & 7u;
            u32 d = 0;
            if (np) {
              d = 0xFFFFu;
              for (u32 x = 0; x < np; ++x) {
                u32 const p = pred_row(g, r, inf, x);
                u32 const hp = g.runhead[p];
                d = min(d, static_cast<u32>(g.rowdepth[hp]) + (p - hp));
              }
              d += 1;
            }
            g.rowdepth[r] = static_cast<u16>(d);
          }
        }
        __syncthreads();
        for (u32 i = lo; i < hi2; ++i) {
          u32 const hp = g.runhead[i];
          if (hp != i) g.rowdepth[i] = static_cast<u16>(g.rowdepth[hp] + (i - hp));
        }
      }
      __syncthreads();
      if (ST.band) band_flags(g, V, ST.band, tid);
      if (ST.overflow) continue;
      PROF_ACC(0);
      if (ws.split && !A.finish && ST.band && ST.nslow <= kSlowCap) {  // hand the fill to k_msa_band<tier> and come back afterwards
        if (tid == 0) {
          ST.c_cur = c;
          ST.h_cur = h;
          ST.pending = 1;
          ST.filled = 0;
          ST.band_fail = 0;
          atomicAdd(ws.pending_ctr, 1u);
          ST.nvars = nvars;
          ST.pool = pool;
          ST.var_overflow = overflow ? 1u : 0u;
        }
        yielded = true;
        break;
      }
      }  // !rh
      // Tiers: the narrow band first (ws.tier0: 64 or 128 columns), then 256 columns, then the full row-synchronous
      // fill.  A band is exact when its certificate holds (see poa_fill_band) and the traceback stays inside it; a
      // tier that fails hands the alignment to the next one -- through k_msa_band again while the batch is in split
      // rounds, inside this kernel (wave 0) in the last launch.
      bool filled = rh && ST.filled;  // k_msa_band has filled tier ST.band between the two launches
      bool const can_yield = ws.split && !A.finish && ST.nslow <= kSlowCap;
      while (true) {
        u32 tier = ST.band;
        if (!filled && tier != kTierIn) tier = 0;  // inside this kernel: the narrow tier or the full fill
        if (!filled) {
          if (ws.dstats && tid == 0) {
            if (tier) {
              atomicAdd(&ws.dstats[7], static_cast<unsigned long long>(V) * (64u * tier));
              atomicAdd(&ws.dstats[8], 1ull);
            } else {
              atomicAdd(&ws.dstats[9], static_cast<unsigned long long>(V) * L);
            }
          }
          if (tier) {
            if (tid == 0) ST.band_fail = 0;
            if (wave == 0) {
              if (ws.lean)
                poa_fill_lean<static_cast<int>(kTierIn)>(g, ws.band_stride, ws.code_cells, codes, rows, hlast, V, L, lane, seq,
                                                         ST.edge0, g.aln.off, &ST.edge_max);
              else
                poa_fill_band<static_cast<int>(kTierIn)>(g, ws.band_stride, ws.code_cells, codes, rows, hlast, V, L, lane, seq,
                                                         &ST.edge_max);
            }
          } else {
            if (cw == 4) {
              poa_fill<4>(g, ws, fcodes, frows, hlast, V, L, tid, seq);
            } else if constexpr (CWMAX > 4) {
              if (cw == 6) {
                poa_fill<6>(g, ws, fcodes, frows, hlast, V, L, tid, seq);
              } else if (cw == 8) {
                poa_fill<8>(g, ws, fcodes, frows, hlast, V, L, tid, seq);
              } else if constexpr (CWMAX > 8) {
                if (cw == 12) poa_fill<12>(g, ws, fcodes, frows, hlast, V, L, tid, seq);
                else poa_fill<16>(g, ws, fcodes, frows, hlast, V, L, tid, seq);
              }
            }
          }
        }
        filled = false;
        __syncthreads();  // full fence: codes and hlast are read below
        PROF_ACC(1);
      // best end cell: first maximum, in rank order, over the nodes without out-edges
        {
          i32 bv = kNegInf;
          u32 br = 0xFFFFFFFFu;
          for (u32 r = tid; r < V; r += kT) {
            if (g.nout[g.rank2node[r]] != 0) continue;
            i32 const hv = hlast[r + 1];
            if (br == 0xFFFFFFFFu || hv > bv) {
              bv = hv;
              br = r + 1;
            }
          }
          for (int off = 32; off > 0; off >>= 1) {
            i32 const ov = __shfl_xor(bv, off);
            u32 const orow = __shfl_xor(br, off);
            if (orow != 0xFFFFFFFFu && (br == 0xFFFFFFFFu || ov > bv || (ov == bv && orow < br))) {
              bv = ov;
              br = orow;
            }
          }
          if (lane == 0) {
            ST.red_v[wave] = bv;
            ST.red_r[wave] = br;
          }
          __syncthreads();
          if (tid == 0) {
            i32 fv = kNegInf;
            u32 fr = 0xFFFFFFFFu;
            for (int k = 0; k < 4; ++k) {
              i32 const ov = ST.red_v[k];
              u32 const orow = ST.red_r[k];
              if (orow != 0xFFFFFFFFu && (fr == 0xFFFFFFFFu || ov > fv || (ov == fv && orow < fr))) {
                fv = ov;
                fr = orow;
              }
            }
            ST.best = fv;
            ST.best_row = fr;
          }
          __syncthreads();
        }
        bool const certified = !tier || (ST.best_row != 0xFFFFFFFFu && ST.best - 32 > ST.edge_max);
        if (wave == 0 && certified) {
          u32 const br = ST.best_row;
          u32 const naln = poa_traceback(g, tier ? codes : fcodes, tier ? ws.code_cells : ws.fcode_cells, cw, V, L, br == 0xFFFFFFFFu ? 0u : br,
                                         br != 0xFFFFFFFFu, lane, 64u * tier);
          if (lane == 0) ST.naln = naln;
        }
        __syncthreads();
        if (!tier || (certified && !ST.band_fail)) break;  // the band was enough / the full fill is always exact
        // ---- next tier ----
        u32 const next = (tier < 4u && can_yield) ? 4u : 0u;
        if (ws.tier_stats && tid == 0) atomicAdd(&ws.tier_stats[4 + (tier == 4u ? 2 : (tier == 2u ? 1 : 0))], 1u);
        __syncthreads();
        if (tid == 0) {
          ST.band = next;
          ST.band_fail = 0;
        }
        if (next) {
          for (u32 i = 1 + tid; i <= V; i += kT) g.rowj0[i] = band_j0(g.npos[g.rank2node[i - 1]], L, next);
          __syncthreads();
          band_flags(g, V, next, tid);
        }
        __syncthreads();
        if (next) {
          if (tid == 0) {
            ST.c_cur = c;
            ST.h_cur = h;
            ST.pending = 1;
            ST.filled = 0;
            atomicAdd(ws.pending_ctr, 1u);
            atomicAdd(ws.pending_ctr + 1, 1u);  // ... at a RETRIED tier: worth a band round of its own (see launch_msa)
            ST.nvars = nvars;
            ST.pool = pool;
            ST.var_overflow = overflow ? 1u : 0u;
          }
          yielded = true;
          break;
        }
      }
      if (yielded) break;
      }  // !direct
      __syncthreads();
      PROF_ACC(2);
      if (ST.overflow) continue;
      // ---- spoa::Graph::AddAlignment; the path is stored reversed in aln ----
      u32 const naln = ST.naln;
      if (naln == 0) {
        if (tid == 0) {
          i32 const first = pg_add_sequence(g, seq, 0, L);
          ST.seq_first[ST.nseq] = first;
          ST.nseq++;
        }
      } else {
        u32 const lab = 1u << ST.nseq;
        // entries that carry a sequence position, compacted in path order
        u32 nv = 0;
        {
          u32 const per = (naln + kT - 1) / kT;
          u32 const lo = min(static_cast<u32>(tid) * per, naln), hi2 = min(lo + per, naln);
          u32 cnt = 0;
          for (u32 x = lo; x < hi2; ++x) cnt += g.aln[2 * (naln - 1 - x) + 1] != 0;
          u32 at = block_excl_scan(cnt, tid, nv);
          for (u32 x = lo; x < hi2; ++x) {
            u32 const p1 = g.aln[2 * (naln - 1 - x) + 1];
            if (p1 == 0) continue;
            g.cnode[at] = g.aln[2 * (naln - 1 - x)];
            g.cpos[at] = static_cast<u16>(p1 - 1);
            ++at;
          }
        }
        __syncthreads();
        if (tid == 0) {
          if (nv == 0) {
            ST.overflow = 1;  // cannot happen for a global alignment of a non-empty sequence
          } else {
            u32 const vfront = g.cpos[0], vback = g.cpos[nv - 1];
            i32 const begin = pg_add_sequence(g, seq, 0, vfront);
            ST.begin = begin;
            ST.prev0 = begin >= 0 ? static_cast<i32>(ST.nn - 1) : -1;
            ST.lastn = pg_add_sequence(g, seq, vback + 1, L);
          }
          ST.nv = nv;
        }
        __syncthreads();
        if (!ST.overflow) {
          u32 const pv = (nv + kT - 1) / kT;  // <= 16
          u32 const vlo = min(static_cast<u32>(tid) * pv, nv), vhi = min(vlo + pv, nv);
          u32 newmask = 0, grpmask = 0;
          for (u32 v = vlo; v < vhi; ++v) {
            u8 const ch = seq[g.cpos[v]];
            u32 const jt1 = g.cnode[v];
            u32 cur = 0xFFFFu;
            if (jt1 != 0) {
              u32 const jt = jt1 - 1;
              if (g.nchar[jt] == ch) {
                cur = jt;
              } else {
                u32 const na = g.nal[jt];
                for (u32 y = 0; y < na; ++y) {
                  u32 const kt = g.al[jt * kPE + y];
                  if (g.nchar[kt] == ch) {
                    cur = kt;
                    break;
                  }
                }
                if (cur == 0xFFFFu) grpmask |= 1u << (v - vlo);
              }
            }
            if (cur == 0xFFFFu) newmask |= 1u << (v - vlo);
            g.ccur[v] = static_cast<u16>(cur);
          }
          u32 total_new = 0;
          u32 id = ST.nn + block_excl_scan(__popc(newmask), tid, total_new);
          bool const fits = ST.nn + total_new <= PN;
          if (fits) {
            for (u32 v = vlo; v < vhi; ++v) {
              if (!(newmask & (1u << (v - vlo)))) continue;
              u32 const cur = id++;
              g.ccur[v] = static_cast<u16>(cur);
              g.nchar[cur] = seq[g.cpos[v]];
              g.nin[cur] = g.nout[cur] = g.nal[cur] = 0;
              {
                // band guide: a node aligned to graph node jt sits where jt sits; an unaligned (inserted) one
                // takes the coordinate of the nearest aligned path entry before it (+ its distance), else v
                u32 pos = v;
                if (g.cnode[v] != 0) {
                  pos = g.npos[static_cast<u32>(g.cnode[v]) - 1];
                } else {
                  for (u32 back = 1; back <= v && back <= 64; ++back)
                    if (g.cnode[v - back] != 0) {
                      pos = static_cast<u32>(g.npos[static_cast<u32>(g.cnode[v - back]) - 1]) + back;
                      break;
                    }
                }
                g.npos[cur] = static_cast<u16>(min(pos, 0xFFFFu));
              }
              if (grpmask & (1u << (v - vlo))) {  // join the aligned ring of node jt
                u32 const jt = static_cast<u32>(g.cnode[v]) - 1;
                u32 const na = g.nal[jt];
                if (na + 1 > kPE) {
                  atomicOr(&ST.overflow, 1u);
                  continue;
                }
                for (u32 y = 0; y < na; ++y) {
                  u32 const kt = g.al[jt * kPE + y];
                  u32 const nk = g.nal[kt];
                  g.al[kt * kPE + nk] = static_cast<u16>(cur);
                  g.nal[kt] = static_cast<u8>(nk + 1);
                  g.al[cur * kPE + y] = static_cast<u16>(kt);
                }
                g.al[jt * kPE + na] = static_cast<u16>(cur);
                g.nal[jt] = static_cast<u8>(na + 1);
                g.al[cur * kPE + na] = static_cast<u16>(jt);
                g.nal[cur] = static_cast<u8>(na + 1);
              }
            }
          }
          __syncthreads();
          if (tid == 0) {
            if (!fits) ST.overflow = 1; else ST.nn += total_new;
          }
          if (fits) {
            for (u32 v = vlo; v < vhi; ++v) {
              i32 const tail = v == 0 ? ST.prev0 : static_cast<i32>(g.ccur[v - 1]);
              if (tail >= 0) pg_add_edge(g, static_cast<u32>(tail), g.ccur[v], lab);
            }
          }
          __syncthreads();
          if (tid == 0 && !ST.overflow) {
            if (ST.lastn >= 0) pg_add_edge(g, g.ccur[nv - 1], static_cast<u32>(ST.lastn), lab);
            ST.seq_first[ST.nseq] = ST.begin >= 0 ? ST.begin : static_cast<i32>(g.ccur[0]);
            ST.nseq++;
          }
        }
      }
      __syncthreads();
      PROF_ACC(3);
      if (!ST.overflow) {
        u32 const nn2 = ST.nn;
        for (u32 i = tid; i < nn2; i += kT) g.marks[i] = g.ignored[i] = 0;
      }
      __syncthreads();
      if (wave == 0 && !ST.overflow) pg_toposort(g, lane);
      __syncthreads();
      if (!ST.overflow) {
        u32 const nrank = ST.nrank;
        for (u32 r = tid; r < nrank; r += kT) g.node2rank[g.rank2node[r]] = static_cast<u16>(r);
      }
      PROF_ACC(4);
    }
    if (yielded) break;
    __syncthreads();

    // ---- VariantExtractor over the component's POA graph: thread 0 walks, everybody helps skipping ----
    bool const do_extract = !ST.overflow && ST.nseq >= 2;
    if (tid == 0 && ST.overflow) overflow = true;
    u32 const ns = ST.nseq;
    u32 const nn = ST.nn;
    i32(&active)[kMaxSeq] = ST.xa_active;
    u32(&hap_pos)[kMaxSeq] = ST.xa_hap_pos;
    u32(&starts)[kMaxSeq] = ST.xa_starts;
    u32 ref_pos = 0;
    i32 prev_match = -1;
    bool x_init = false;
    u32 const allmask = ns >= 32 ? 0xFFFFFFFFu : ((1u << ns) - 1u);
    while (do_extract) {
      if (tid == 0) {
        if (!x_init) {
          for (u32 s = 0; s < ns; ++s) {
            active[s] = ST.seq_first[s];
            hap_pos[s] = 0;
          }
          ref_pos = A.a.comp_anchor[ci];  // window-relative ref_anchor_pos (variant_builder.cpp:146)
          x_init = true;
        } else {
          // converged at node a: the next `run` nodes a, a+1, ... are passed by every haplotype
          u32 run = ST.wsum[0];
          if (run == 64) {
            run += ST.wsum[1];
            if (run == 128) {
              run += ST.wsum[2];
              if (run == 192) run += ST.wsum[3];
            }
          }
          if (run == 0) {  // one plain converged step
            prev_match = active[0];
            for (u32 s = 0; s < ns; ++s)
              if (active[s] >= 0) {
                active[s] = pg_successor(g, static_cast<u32>(active[s]), s);
                hap_pos[s]++;
              }
            ref_pos++;
          } else {
            i32 const a = active[0];
            prev_match = a + static_cast<i32>(run) - 1;
            for (u32 s = 0; s < ns; ++s) {
              active[s] = a + static_cast<i32>(run);
              hap_pos[s] += run;
            }
            ref_pos += run;
          }
        }
        u8* pl = A.o.allele_pool + static_cast<size_t>(w) * MP;
        u32 const acap_hbm = 2 * ws.max_l + 8;  // raw allele strings: [ns][acap]
        u32(&rawlen)[kMaxSeq] = ST.xa_rawlen;
        auto converged = [&]() {
          for (u32 s = 1; s < ns; ++s)
            if (active[s] != active[0]) return false;
          return true;
        };
        bool done = false;
        while (true) {
          if (converged()) {
            if (active[0] < 0) done = true;
            break;  // hand the converged stretch to the whole workgroup
          }
          bool const has_prev = prev_match >= 0;
          u32 const aoff = has_prev ? 1u : 0u;
          u32 start_pos = ref_pos - aoff;
          u8* raw = raw_lds;  // this bubble's strings: in LDS until one outgrows its share
          u32 acap = raw_lds_cap;
          for (u32 s = 0; s < ns; ++s) {
            rawlen[s] = 0;
            if (has_prev) raw[s * acap + rawlen[s]++] = g.nchar[prev_match];
            starts[s] = hap_pos[s] - aoff;
          }
          while (!converged()) {
            u32 min_rank = 0xFFFFFFFFu;
            for (u32 s = 0; s < ns; ++s)
              if (active[s] >= 0) min_rank = min(min_rank, static_cast<u32>(g.node2rank[active[s]]));
            if (min_rank == 0xFFFFFFFFu) break;
            for (u32 s = 0; s < ns; ++s)
              if (active[s] >= 0 && g.node2rank[active[s]] == min_rank) {
                if (rawlen[s] >= acap && raw == raw_lds) {  // move the bubble to HBM
                  for (u32 t = 0; t < ns; ++t)
                    for (u32 x = 0; x < rawlen[t]; ++x) raw_hbm[t * acap_hbm + x] = raw_lds[t * raw_lds_cap + x];
                  raw = raw_hbm;
                  acap = acap_hbm;
                }
                if (rawlen[s] < acap) raw[s * acap + rawlen[s]++] = g.nchar[active[s]]; else overflow = true;
                active[s] = pg_successor(g, static_cast<u32>(active[s]), s);
                hap_pos[s]++;
                if (s == 0) ref_pos++;
              }
          }
          // group identical non-REF alleles (CreateNormalizedBubble); alt_of[s] = group id or -1
          i32(&alt_of)[kMaxSeq] = ST.xa_alt_of;
          u32(&grp_rep)[kMaxSeq] = ST.xa_grp_rep;
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
          u32(&glo)[kMaxSeq] = ST.xa_glo;
          u32(&ghi)[kMaxSeq] = ST.xa_ghi;
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
          u32(&ordg)[kMaxSeq] = ST.xa_ordg;
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
            done = true;
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
          u32(&rank_of_grp)[kMaxSeq] = ST.xa_rank_of_grp;
          for (u32 ai = 0; ai < ngrp; ++ai) {
            u32 const gi = ordg[ai];
            rank_of_grp[gi] = ai;
            const u8* as = raw + grp_rep[gi] * acap + glo[gi];
            u32 const alen = ghi[gi] - glo[gi];
            A.o.alt_off[vi * MA + ai] = pool;
            A.o.alt_len[vi * MA + ai] = alen;
            for (u32 x = 0; x < alen; ++x) pl[pool++] = as[x];
            const u8* rs = raw + rlo;  // (the REF allele as it stands in the scratch: the pool copy is in HBM)
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
        ST.x_done = done ? 1u : 0u;
        ST.x_a = static_cast<u32>(active[0]);
      }
      __syncthreads();
      if (ST.x_done) break;
      {
        // thread k: do ALL haplotypes step from node a + k to node a + k + 1?
        u32 const v = ST.x_a + static_cast<u32>(tid);
        bool ok = false;
        if (v + 1 < nn) {
          u32 const no = g.nout[v];
          for (u32 x = 0; x < no; ++x)
            if (g.out_head[v * kPE + x] == v + 1 && (g.lab(v * kPE + x) & allmask) == allmask) ok = true;
        }
        unsigned long long const m = __ballot(ok);
        u32 const cnt = m == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~m));
        if (lane == 0) ST.wsum[wave] = cnt;
      }
      __syncthreads();
    }
    if (tid == 0 && overflow) ST.win_overflow = 1;
    PROF_ACC(5);
  }
  if (yielded) {  // save the LDS block; the next launch of this kernel resumes from it
    __syncthreads();
    {
      uint4* dst = reinterpret_cast<uint4*>(img);
      const uint4* src = reinterpret_cast<const uint4*>(ma_lds);
      u32 const nq = min(ws.img_words / 4u, (g.aln.off + 15u) / 16u);
      for (u32 i = tid; i < nq; i += kT) dst[i] = src[i];
    }
    PROF_ACC(14);  // image save
    // (k_poa publishes the window to another workgroup right after this: every wavefront's stores have to be out of its
    //  own queue before thread 0's device-scope release -- a workgroup-scope barrier alone does not wait for them)
    __builtin_amdgcn_s_waitcnt(0);
    return kMsaYield;
  }
  if (tid == 0) {
    A.o.win_nvars[w] = nvars;
    if (overflow) A.a.win_status[w] |= MA_W_VAR_OVERFLOW;
    if (ws.split) {
      ST.done = 1;
      ST.pending = 0;
    }
  }
  if (ws.split) {
    __syncthreads();
    for (u32 i = tid; i < kStBytes / 4; i += kT) img[i] = reinterpret_cast<const u32*>(ma_lds)[i];
  }
  return kMsaDone;
}

template <int CWMAX, bool LAB32>
__global__ __launch_bounds__(kT, LAB32 ? 1 : 2) void k_msa(MsaArgs A) {
  (void)msa_window<CWMAX, LAB32>(A, static_cast<int>(blockIdx.x), A.round > 0);
}

// The row descriptors of a pending alignment, read straight from the window's LDS image in HBM: every access is
// wave-uniform and the fill asks for row i + 1 while it works on row i, so no LDS staging is needed and the
// occupancy of k_msa_band is bounded by its registers alone.
// Every read is a SCALAR load (constant address space -> s_load_dword, counted by lgkmcnt): a vector load would share
// the in-order vmcnt counter with the row's code store, and waiting for a descriptor would mean waiting for the store
// issued just before it -- a store round trip (~1 us under load) on the critical path of EVERY row.
#define MA_AS4 __attribute__((address_space(4)))
__device__ __forceinline__ u32 sload_u32(const void* p) {  // p uniform, 4-byte aligned
  return *reinterpret_cast<const MA_AS4 u32*>(reinterpret_cast<uintptr_t>(p));
}
template <class T>
struct ScalarArr {
  const T* base;
  __device__ __forceinline__ u32 operator[](u32 i) const {
    uintptr_t const a = reinterpret_cast<uintptr_t>(base + i);
    u32 const w = sload_u32(reinterpret_cast<const void*>(a & ~uintptr_t(3)));
    if constexpr (sizeof(T) == 4) return w;
    else if constexpr (sizeof(T) == 2) return (w >> ((a & 2u) * 8u)) & 0xFFFFu;
    else return (w >> ((a & 3u) * 8u)) & 0xFFu;
  }
};
struct DescView {
  ScalarArr<u32> rowinfo;
  ScalarArr<u16> rowslot, rowdepth, rowj0, slowpred;
  // only reached for rows without a cached predecessor list, which k_msa never hands over (nslow <= kSlowCap)
  ScalarArr<u16> rank2node, node2rank, in_tail;
  // one byte at a wave-uniform address
  __device__ __forceinline__ u32 ubyte(const u8* p) const { return ScalarArr<u8>{p}[0]; }
  __device__ __forceinline__ u32 uword(const void* p) const { return sload_u32(p); }
};

// Split mode: the banded fill (tier CW: 64 CW columns) of every window whose pending alignment is at that tier, one
// wavefront per window.  Only the state block is staged in LDS (instead of the graph's 77 KB), so a CU holds 16-32
// windows and the dependent instruction chains of the row recurrence overlap across them.
// One launch serves every tier: the wavefront takes the tier its window waits at (a launch per tier cost each tier's
// slowest window in turn -- and a tier with three windows still cost a fill's latency).
// The banded fill of window lw's pending alignment by ONE wavefront (`lane` = its lane index): the body of k_msa_band
// (host-counted rounds) and the F job of the persistent kernel k_poa.  Nothing of the window is staged in LDS -- the state
// block's few fields and the row descriptors come straight from the image in HBM -- so any wavefront of any workgroup can
// run it; lut_off = 256 bytes of LDS of the wavefront's own for poa_fill_lean's code table.
template <bool LEAN>
__device__ __forceinline__ bool band_job(MsaArgs const& A, int const lw, int const lane, u32 const lut_off) {
  int const w = A.win0 + lw;
  ma_params_t const& P = A.prm;
  PoaWs const& ws = A.ws;
  u32* const img = reinterpret_cast<u32*>(ws.img) + static_cast<size_t>(lw) * ws.img_words;
  WgState* const pst = reinterpret_cast<WgState*>(img);
  if (pst->done || !pst->pending || pst->filled || pst->band == 0) return false;
  u32 const PN = ws.pn;
  GL const full = poa_carve(PN, ws.max_l, ws.lab32);
  const u8* const ib = reinterpret_cast<const u8*>(img);
  DescView g;
  g.rowinfo.base = reinterpret_cast<const u32*>(ib + full.rowinfo.off);
  g.rowslot.base = reinterpret_cast<const u16*>(ib + full.rowslot.off);
  g.rowdepth.base = reinterpret_cast<const u16*>(ib + full.rowdepth.off);
  g.rowj0.base = reinterpret_cast<const u16*>(ib + full.rowj0.off);
  g.slowpred.base = reinterpret_cast<const u16*>(ib + full.slowpred.off);
  g.rank2node.base = reinterpret_cast<const u16*>(ib + full.rank2node.off);
  g.node2rank.base = reinterpret_cast<const u16*>(ib + full.node2rank.off);
  g.in_tail.base = reinterpret_cast<const u16*>(ib + full.in_tail.off);
  // (what comes out of a vector load counts as divergent: without the readfirstlanes the haplotype's address sits in vector
  //  registers and every "scalar" load of the fill turns into a vector load)
  u32 const c_cur = __builtin_amdgcn_readfirstlane(pst->c_cur), h_cur = __builtin_amdgcn_readfirstlane(pst->h_cur);
  u32 const V = __builtin_amdgcn_readfirstlane(pst->V), L = __builtin_amdgcn_readfirstlane(pst->L);
  i32 const edge0 = __builtin_amdgcn_readfirstlane(pst->edge0);
  size_t const ci = static_cast<size_t>(w) * P.max_comps + c_cur;
  u32 const hap0 = __builtin_amdgcn_readfirstlane(A.a.comp_hap0[ci]);
  size_t const hi = static_cast<size_t>(w) * P.max_haps + hap0 + h_cur;
  const u8* seq = A.a.hap_bases + hi * P.max_hap_len;
  u16* const codes = reinterpret_cast<u16*>(reinterpret_cast<u8*>(ws.codes) + static_cast<size_t>(lw) * ws.code_cells * 2);
  i32* const rows = ws.rows + static_cast<size_t>(lw) * ws.row_cells;
  i32* const hlast = ws.hlast + static_cast<size_t>(lw) * (PN + 8);
  u32 const tier = __builtin_amdgcn_readfirstlane(pst->band);
  auto const fill = [&](auto cw) {
    constexpr int CW = decltype(cw)::value;
    if constexpr (LEAN)
      poa_fill_lean<CW>(g, ws.band_stride, ws.code_cells, codes, rows, hlast, V, L, lane, seq, edge0, lut_off, &pst->edge_max);
    else
      poa_fill_band<CW>(g, ws.band_stride, ws.code_cells, codes, rows, hlast, V, L, lane, seq, &pst->edge_max);
  };
  if (tier == 1) fill(std::integral_constant<int, 1>{});
  else if (tier == 2) fill(std::integral_constant<int, 2>{});
  else fill(std::integral_constant<int, 4>{});
  if (lane == 0) {
    pst->filled = 1;
    if (ws.tier_stats) atomicAdd(&ws.tier_stats[tier == 1 ? 0 : (tier == 2 ? 1 : 2)], 1u);
    if (ws.dstats) {
      atomicAdd(&ws.dstats[7], static_cast<unsigned long long>(V) * (64u * tier));
      atomicAdd(&ws.dstats[8], 1ull);
    }
  }
  return true;
}

template <bool LEAN>
__global__ __launch_bounds__(64, 4) void k_msa_band(MsaArgs A) {
  int const lw = blockIdx.x;
  if (poa_window_skipped(A, A.win0 + lw)) return;
  (void)band_job<LEAN>(A, lw, static_cast<int>(threadIdx.x), 16u);
}

// ---- the persistent POA kernel: device-side scheduling instead of host-counted rounds (round 6) ----------------------------
// Rounds 3-5 ran the POA as rounds the HOST counted: k_msa over every window of the chunk (restore the LDS image, work up to the
// next fill, save), one 8-byte read-back, k_msa_band over every window, again -- 17 + 12 launches and ~30 stream
// synchronisations per lane-step, every round as long as its slowest window, the later rounds (third / fourth alignments,
// retried tiers) nearly empty but a fill's latency each.  k_poa is ONE launch per chunk: its workgroups loop over queues in HBM
// until every window is done --
//   G jobs (a whole workgroup, the window's POA graph in LDS): msa_window up to the window's next banded fill, which it posts
//          as an F job (fresh windows come off a counter, windows whose fill is done off the G queue);
//   F jobs (ONE wavefront each, up to four side by side in a workgroup): band_job, then the window goes back on the G queue.
// Fills are taken first (they are the long jobs: the graph phases of other windows run while they are in flight).
// Windows hand over through HBM (the LDS image, the decision codes).  **Every hand-over stays inside one XCD**: the chip's
// eight XCDs have an L2 each, and a device-scope release / acquire on this chip is a write-back of the whole L2 plus an
// invalidation of it (buffer_wbl2 sc1 / buffer_inv sc1) -- the first version, with __threadfence() around every queue
// operation, ran the stage 2.6x SLOWER than the host rounds and slowed the other lanes' kernels with it.  A window belongs to
// the XCD of the workgroup that started it (s_getreg XCC_ID): its F and G jobs go through that XCD's queues and are taken by
// that XCD's workgroups only, so producer and consumer share one coherent L2 and a hand-over costs: the producer's stores out
// of its queues (s_waitcnt vmcnt(0), every wavefront, before the slot is published), the consumer's vector L1 invalidated
// (buffer_inv sc0) -- and its SCALAR cache (s_dcache_inv), which the fill's descriptor loads go through and no fence covers.
// Fresh windows come off one global counter, so the XCDs balance themselves.  A workgroup leaves when no window is left to
// start and its XCD has more workgroups than open windows.  (MA_POA_XCD=0: one domain for the whole chip, device-scope fences.)
constexpr u32 kPoaDoms = 16;
struct PoaDom {  // one per XCD, a cache line of its own
  u32 f_head, f_tail, g_head, g_tail;
  u32 h_head, h_tail;  // fills of windows that are past their first alignment: taken first (the batch's critical path)
  u32 n_open;     // windows started here and not finished
  u32 n_workers;  // workgroups of this XCD in the loop
  u32 seen;       // workgroups that ever registered here, windows started here (MA_VERBOSE prints them)
  u32 started;
  u32 pad[22];
};
struct PoaSched {
  u32 fresh;  // next window that has not been started
  u32 pad[31];
  PoaDom dom[kPoaDoms];
};
constexpr u32 kQEmpty = 0xFFFFFFFFu;
__device__ __forceinline__ u32 ld_dev(u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// multi-producer / multi-consumer ring: a slot is reserved by the counter, published by its value (kQEmpty = not yet / taken).
// The caller has made the job's data visible BEFORE the push (poa_release) and makes it readable AFTER the pop (poa_acquire).
__device__ __forceinline__ void q_push(u32* tail, u32* buf, u32 mask, u32 id) {
  u32 const t = atomicAdd(tail, 1u);
  u32* const slot = buf + (t & mask);
  while (ld_dev(slot) != kQEmpty) __builtin_amdgcn_s_sleep(2);  // (the ring holds every window at once: practically never)
  st_dev(slot, id);
}
__device__ __forceinline__ u32 q_pop(u32* head, u32* tail, u32* buf, u32 mask, u32 maxn, u32* out) {
  for (;;) {
    u32 const h = ld_dev(head), t = ld_dev(tail);
    if (static_cast<i32>(t - h) <= 0) return 0;
    u32 const n = min(maxn, t - h);
    if (atomicCAS(head, h, h + n) != h) continue;
    for (u32 k = 0; k < n; ++k) {
      u32* const slot = buf + ((h + k) & mask);
      u32 v;
      while ((v = ld_dev(slot)) == kQEmpty) __builtin_amdgcn_s_sleep(1);  // reserved, about to be published
      out[k] = v;
      st_dev(slot, kQEmpty);
    }
    return n;
  }
}
// producer side, EVERY wavefront that wrote: its stores have left its queues (they are in the XCD's L2) ...
__device__ __forceinline__ void poa_release(bool device_scope) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  if (device_scope) __threadfence();  // ... and, with one domain for the whole chip, written back for the other XCDs
}
// consumer side, every wavefront that is going to read: nothing stale in the vector L1 or in the scalar cache
__device__ __forceinline__ void poa_acquire(bool device_scope) {
  if (device_scope) __threadfence();
  asm volatile("buffer_inv sc0\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The order in which k_poa starts a chunk's windows: most alignments first.  A launch lasts at least as long as its longest
// window -- alignments are a chain (graph phase, fill, graph phase, ...) -- and a window with three of them that starts when the
// others are through holds the launch, and every CU's LDS, for itself.  Counting sort by the number of alignments (33 buckets;
// the order inside a bucket does not matter: windows are independent, every window's result is its own).
__global__ __launch_bounds__(1024) void k_poa_order(MsaArgs A, u32* order, u32 nwin) {
  __shared__ u32 cnt[34];
  if (threadIdx.x < 34) cnt[threadIdx.x] = 0;
  __syncthreads();
  auto bucket_of = [&](u32 lw) {
    int const w = A.win0 + static_cast<int>(lw);
    if (poa_window_skipped(A, w)) return 33u;  // (returns at once: last)
    u32 al = 0;
    u32 const nc = A.a.win_ncomp[w];
    for (u32 c = 0; c < nc; ++c) {
      u32 const nh = A.a.comp_nhaps[static_cast<size_t>(w) * A.prm.max_comps + c];
      al += nh > 1 ? nh - 1 : 0;
    }
    return 32u - min(al, 32u);
  };
  for (u32 lw = threadIdx.x; lw < nwin; lw += 1024) atomicAdd(&cnt[bucket_of(lw)], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (u32 x = 0; x < 34; ++x) {
      u32 const c = cnt[x];
      cnt[x] = run;
      run += c;
    }
  }
  __syncthreads();
  for (u32 lw = threadIdx.x; lw < nwin; lw += 1024) order[atomicAdd(&cnt[bucket_of(lw)], 1u)] = lw;
}

template <int CWMAX, bool LAB32>
__global__ __launch_bounds__(kT, LAB32 ? 1 : 2) void k_poa(MsaArgs A, PoaSched* S, u32* fq_all, u32* gq_all, u32* hq_all, u32 qcap, u32 nwin,
                                                           u32 xcd_local, u32 policy_min_fills, const u32* order) {
  __shared__ u32 sh_job[8];  // [0] kind: 0 nothing right now, 1 fresh window, 2 window back from its fill, 3 fills, 4 leave; [1] count; [2..5] windows
  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // HW_REG_XCC_ID (id 20), bits [3:0]: the XCD this workgroup runs on
  u32 const dom_id = xcd_local ? (static_cast<u32>(__builtin_amdgcn_s_getreg(20 | (3 << 11))) & (kPoaDoms - 1u)) : 0u;
  PoaDom* const D = &S->dom[dom_id];
  u32* const fq = fq_all + static_cast<size_t>(dom_id) * qcap;
  u32* const gq = gq_all + static_cast<size_t>(dom_id) * qcap;
  u32* const hq = hq_all + static_cast<size_t>(dom_id) * qcap;
  u32 const qmask = qcap - 1u;
  bool const dev_scope = !xcd_local;
  if (tid == 0) {
    atomicAdd(&D->n_workers, 1u);
    atomicAdd(&D->seen, 1u);
  }
#ifdef MA_PROFILE
  unsigned long long pt = __builtin_amdgcn_s_memtime();
  unsigned long long const pt_begin = pt;
#define KPOA_ACC(slot) do { unsigned long long const _n = __builtin_amdgcn_s_memtime(); if (tid == 0) atomicAdd(&g_prof2[slot], _n - pt); pt = _n; } while (0)
#else
#define KPOA_ACC(slot) do {} while (0)
#endif
  for (;;) {
    __syncthreads();  // (the previous job's last LDS reads)
    if (tid == 0) {
      u32 kind = 0, cnt = 0, ids[4] = {0, 0, 0, 0};
      // A fill occupies its wavefront for ~0.5 ms and the whole workgroup waits for it: fills are taken when there is a
      // workgroup's worth of them (four) -- or nothing else to do; until then windows that came back from their fill go first
      // (they are the furthest along: the batch's critical path is its windows with three or four alignments), then fresh ones.
      auto pop_fills = [&]() {  // later alignments first
        u32 n = q_pop(&D->h_head, &D->h_tail, hq, qmask, 4u, ids);
        if (n < 4u) n += q_pop(&D->f_head, &D->f_tail, fq, qmask, 4u - n, ids + n);
        return n;
      };
      u32 const f_avail = (ld_dev(&D->f_tail) - ld_dev(&D->f_head)) + (ld_dev(&D->h_tail) - ld_dev(&D->h_head));
      if (static_cast<i32>(f_avail) >= static_cast<i32>(policy_min_fills)) cnt = pop_fills();
      if (cnt) {
        kind = 3;
      } else if (q_pop(&D->g_head, &D->g_tail, gq, qmask, 1u, ids)) {
        kind = 2;
        cnt = 1;
      } else if (ld_dev(&S->fresh) >= nwin && (cnt = pop_fills()) != 0) {
        kind = 3;  // (no window left to start: whatever fills there are)
      } else {
        bool more = ld_dev(&S->fresh) < nwin;
        if (more) {
          u32 const f = atomicAdd(&S->fresh, 1u);
          if (f < nwin) {
            kind = 1;
            cnt = 1;
            ids[0] = order ? order[f] : f;  // (k_poa_order: the windows with the most alignments start first)
            atomicAdd(&D->n_open, 1u);
            atomicAdd(&D->started, 1u);
          } else {
            more = false;
          }
        }
        if (!kind && !more) {  // no window left to start: leave once this XCD has more workgroups than open windows
          u32 const open = ld_dev(&D->n_open);
          if (ld_dev(&D->n_workers) > open) {
            if (atomicSub(&D->n_workers, 1u) > open) kind = 4; else atomicAdd(&D->n_workers, 1u);
          }
        }
      }
      sh_job[0] = kind;
      sh_job[1] = cnt;
      for (u32 k = 0; k < 4; ++k) sh_job[2 + k] = ids[k];
    }
    __syncthreads();
    u32 const kind = sh_job[0], cnt = sh_job[1];
    KPOA_ACC(0);
    if (kind == 4) break;
    if (kind == 0) {
      __builtin_amdgcn_s_sleep(64);
      KPOA_ACC(3);
#ifdef MA_PROFILE
      if (tid == 0) atomicAdd(&g_prof2[9], 1ull);
#endif
      continue;
    }
    if (kind == 3) {
      if (static_cast<u32>(wave) < cnt) {
        int const lw = static_cast<int>(__builtin_amdgcn_readfirstlane(sh_job[2 + wave]));
        poa_acquire(dev_scope);  // the image another workgroup saved
        (void)band_job<true>(A, lw, lane, static_cast<u32>(wave) * 256u);
        poa_release(dev_scope);  // codes, stored rows, last column, edge maximum, the image's `filled`
        if (lane == 0) q_push(&D->g_tail, gq, qmask, static_cast<u32>(lw));
      }
#ifdef MA_PROFILE
      __syncthreads();
      KPOA_ACC(2);
      if (tid == 0) atomicAdd(&g_prof2[3 + cnt], 1ull);
#endif
      continue;
    }
    int const lw = static_cast<int>(sh_job[2]);
    if (kind == 2) poa_acquire(dev_scope);  // the fill's outputs
    u32 const st = msa_window<CWMAX, LAB32>(A, lw, kind == 2);
    if (st == kMsaYield) poa_release(dev_scope);  // the image (every wavefront wrote a part)
    __syncthreads();
    if (tid == 0) {
      if (st == kMsaYield) {
        // (the state block is still in LDS: which alignment of the window is this?)
        bool const later = ST.c_cur > 0 || ST.h_cur > 1 || ST.band > A.ws.tier0;
        if (later) q_push(&D->h_tail, hq, qmask, static_cast<u32>(lw)); else q_push(&D->f_tail, fq, qmask, static_cast<u32>(lw));
      } else {
        atomicSub(&D->n_open, 1u);
      }
    }
    KPOA_ACC(1);
#ifdef MA_PROFILE
    if (tid == 0) atomicAdd(&g_prof2[8], 1ull);
#endif
  }
#ifdef MA_PROFILE
  if (tid == 0) atomicAdd(&g_prof2[10], __builtin_amdgcn_s_memtime() - pt_begin);
#endif
}

}  // namespace

#ifdef MA_PROFILE
extern "C" void ma_debug_prof2(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof2), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof2), z, sizeof(z));
  }
}
extern "C" void ma_debug_prof(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
  }
}
#endif

// longest haplotype that is aligned and the most alignments of any window (what launch_msa sizes the graph and its rounds by)
__global__ __launch_bounds__(256) void k_msa_maxima(ma_asm_out_t a, int n, u32 max_haps, u32 max_comps, u32* out) {
  int const w = blockIdx.x * 256 + threadIdx.x;
  // out[0] / out[1]: over all windows; out[2]: most haplotypes of any component; out[3]: longest haplotype of a window
  // that holds such a wide component (sizes the LAB32 pass)
  u32 ml = 0, al = 0, mh = 0, mlw = 0;
  if (w < n && !(a.win_status[w] & MA_W_NO_HAPLOTYPE)) {
    for (u32 c = 0; c < a.win_ncomp[w]; ++c) {
      size_t const ci = static_cast<size_t>(w) * max_comps + c;
      u32 const nh = a.comp_nhaps[ci], h0 = a.comp_hap0[ci];
      for (u32 h = 0; h < nh; ++h) ml = max(ml, a.hap_len[static_cast<size_t>(w) * max_haps + h0 + h]);
      al += nh > 1 ? nh - 1 : 0;
      mh = max(mh, nh);
    }
    if (mh > 16) mlw = ml;
  }
  for (int off = 32; off > 0; off >>= 1) {
    ml = max(ml, __shfl_xor(ml, off));
    al = max(al, __shfl_xor(al, off));
    mh = max(mh, __shfl_xor(mh, off));
    mlw = max(mlw, __shfl_xor(mlw, off));
  }
  if ((threadIdx.x & 63) == 0) {
    if (ml) atomic_max_lazy(&out[0], ml);
    if (al) atomic_max_lazy(&out[1], al);
    if (mh) atomic_max_lazy(&out[2], mh);
    if (mlw) atomic_max_lazy(&out[3], mlw);
  }
}
static int launch_msa_on_stream(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o);
// The POA rounds of a lane are short launches of few, long-lived wavefronts (one per window): next to another lane's
// throughput kernel -- thousands of workgroups queued for every wave slot -- each of their launches waits for slots like
// everybody else, 29 times per lane-step.  They run on a stream of the greatest priority instead: the dispatcher takes
// their workgroups first, which costs the throughput kernels a few slots and takes the POA's chain of launches off the
// step's critical path.  (MA_POA_PRIORITY=0: the lane's own stream, as before.)
int launch_msa(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o) {
  static bool const use_hi = !(getenv("MA_POA_PRIORITY") && atoi(getenv("MA_POA_PRIORITY")) == 0);
  if (!use_hi || b.n_windows == 0) return launch_msa_on_stream(ctx, b, a, o);
  if (!ctx->hi_stream && !ctx->hi_failed) {  // (a device without stream priorities: the lane's own stream, no error)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->hi_stream, hipStreamNonBlocking, greatest) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->hi_ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (ctx->hi_stream) (void)hipStreamDestroy(ctx->hi_stream);
      ctx->hi_stream = nullptr;
      ctx->hi_failed = true;
    }
  }
  if (!ctx->hi_stream) return launch_msa_on_stream(ctx, b, a, o);
  hipStream_t const mine = ctx->stream;
  MA_HIP(ctx, hipEventRecord(ctx->hi_ev, mine));
  MA_HIP(ctx, hipStreamWaitEvent(ctx->hi_stream, ctx->hi_ev, 0));
  ctx->stream = ctx->hi_stream;
  int const rc = launch_msa_on_stream(ctx, b, a, o);
  ctx->stream = mine;
  if (hipEventRecord(ctx->hi_ev, ctx->hi_stream) != hipSuccess || hipStreamWaitEvent(mine, ctx->hi_ev, 0) != hipSuccess) {
    ma_set_err(ctx, "ma_msa_batch: joining the POA stream failed");
    return rc != MA_OK ? rc : MA_ERR_HIP;
  }
  return rc;
}
static int launch_msa_pass(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o, u32 max_len, u32 rounds,
                           u32 lab32, u32 pass);
static int launch_msa_on_stream(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o) {
  int const n = b.n_windows;
  if (n == 0) return MA_OK;
  ma_params_t const& P = ctx->prm;
  if (P.max_haps > static_cast<int>(kMaxSeq)) {
    ma_set_err(ctx, "ma_msa_batch: max_haps > 32 not supported (32-bit haplotype label masks)");
    return MA_ERR_PARAM;
  }
  if (P.max_hap_len > 4096) {
    ma_set_err(ctx, "ma_msa_batch: max_hap_len > 4096 not supported (256 lanes x 16 columns)");
    return MA_ERR_PARAM;
  }
  // longest haplotype of the batch decides the DP width and the LDS graph capacity (one small D2H)
  u32 got[4] = {0, 0, 0, 0};
  {
    // (on the device: the five arrays used to come back whole -- 0.3 MB per lane into pageable memory, staged copies and a
    //  host loop in front of every lane's POA rounds -- for a few numbers)
    MA_HIP(ctx, ctx->ws_build.reserve(4096));
    u32* mx = static_cast<u32*>(ctx->ws_build.p);
    MA_HIP(ctx, hipMemsetAsync(mx, 0, 16, ctx->stream));
    hipLaunchKernelGGL(k_msa_maxima, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a, n, static_cast<u32>(P.max_haps),
                       static_cast<u32>(P.max_comps), mx);
    MA_HIP(ctx, hipMemcpyAsync(got, mx, 16, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, ma_stream_sync(ctx));
  }
  u32 const rounds = std::max<u32>(1u, got[1] + 1u);  // split mode: 1 + the most alignments of any window
  bool const any_wide = got[2] > 16 && !getenv("MA_POA_NO_LAB32");  // (knob for tests: wide components stay flagged, as in round 5)
  if (getenv("MA_POA_FORCE_LAB32")) return launch_msa_pass(ctx, b, a, o, got[0], rounds, 1u, 0u);  // tests: every window through LAB32
  // The reference has no cap on the haplotypes of a component (cbdg/graph.cpp:846-924, caller/msa_builder.cpp:29-42); the
  // engine's is the caller's max_haps <= 32.  Components of up to 16 haplotypes take the common kernels (16-bit label masks,
  // two workgroups per CU); a window with a wider component takes the LAB32 kernels in a pass of its own.
  MA_TRY_RC(launch_msa_pass(ctx, b, a, o, got[0], rounds, 0u, any_wide ? 1u : 0u));
  if (any_wide) MA_TRY_RC(launch_msa_pass(ctx, b, a, o, std::max(got[3], 16u), rounds, 1u, 2u));
  return MA_OK;
}
static int launch_msa_pass(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o, u32 max_len, u32 rounds,
                           u32 const lab32, u32 const pass) {
  int const n = b.n_windows;
  ma_params_t const& P = ctx->prm;
  PoaWs ws{};
  MA_TRY_RC(ma_dev_stats(ctx, &ws.dstats));
  ws.lab32 = lab32;
  ws.pass = pass;
  max_len = std::max<u32>(max_len, 16);
  ws.max_l = max_len;
  u32 pn = max_len + std::max<u32>(256, max_len / 4);
  if (const char* e = getenv("MA_POA_NODE_CAP")) pn = static_cast<u32>(atoi(e));
  pn = std::min<u32>((pn + 7) & ~7u, 65000);
  // two workgroups per CU when the graph fits in 80 KB of LDS, one otherwise
  while (!lab32 && poa_lds_bytes(pn, max_len, lab32) > 80 * 1024 && pn > max_len + 128) pn -= 8;
  while (poa_lds_bytes(pn, max_len, lab32) > 159 * 1024 && pn > max_len + 32) pn -= 8;
  size_t const lds = poa_lds_bytes(pn, max_len, lab32);
  if (lds > 160 * 1024) {
    ma_set_err(ctx, "ma_msa_batch: haplotypes too long for the LDS-resident POA graph");
    return MA_ERR_PARAM;
  }
  ws.pn = pn;
  ws.w_stride = (max_len + 7) & ~7u;
  ws.row_slots = pn / 2;
  // MA_POA_BAND: 2 (default) = 256-column banded fill in its own kernel (k_msa_band), 1 = banded fill inside k_msa,
  // 0 = full row-synchronous fill only.  Results are identical (the band is certified or redone in full).
  // (3, for tests: the persistent kernel with every alignment through the full fill -- its per-workgroup areas)
  int const band_mode = getenv("MA_POA_BAND") ? atoi(getenv("MA_POA_BAND")) : 2;
  ws.use_band = (band_mode != 0 && band_mode != 3) ? 1u : 0u;
  ws.split = band_mode >= 2 ? 1u : 0u;
  ws.no_direct = getenv("MA_POA_NO_DIRECT") ? 1u : 0u;
  ws.raw_cap = getenv("MA_POA_RAW_CAP") ? static_cast<u32>(atoi(getenv("MA_POA_RAW_CAP"))) : 0u;
  ws.lean = (getenv("MA_POA_LEAN") && atoi(getenv("MA_POA_LEAN")) == 0) ? 0u : 1u;
  // first band tier: 64 (1), 128 (2) or 256 (4) columns; a tier whose certificate fails hands over to 256 columns, then
  // to the full fill.  Results do not depend on it (tested).
  {
    int const t0 = getenv("MA_POA_TIER0") ? atoi(getenv("MA_POA_TIER0")) : MA_POA_TIER0;
    ws.tier0 = (t0 == 1 || t0 == 2) ? static_cast<u32>(t0) : 4u;
    ws.no_wide_start = getenv("MA_POA_NO_WIDE_START") ? 1u : 0u;
  }
  // split mode: a band round is launched while at least this many windows wait for a fill; fewer finish inside k_msa
  u32 const min_pending = getenv("MA_POA_MIN_PENDING") ? static_cast<u32>(atoi(getenv("MA_POA_MIN_PENDING"))) : 256u;
  bool const verbose = getenv("MA_VERBOSE") != nullptr;
  bool const no_retry_rounds = getenv("MA_POA_NO_RETRY_ROUNDS") != nullptr;  // (A/B: round 3's rule)
  ws.img_words = static_cast<u32>((lds + 3) / 4);
  // A window's own areas hold what the band tiers write (at most 256 columns per row: 2.7 MB per window of 1 kb haplotypes);
  // the full fill's whole-row areas (10 MB) belong to whoever runs it -- see PoaWs.
  ws.band_stride = 256;
  ws.code_cells = (static_cast<size_t>(pn + 2) * 256 + 7) & ~size_t(7);
  if (ws.code_cells * 2 < static_cast<size_t>(P.max_haps) * (2 * max_len + 8) + 64)  // raw-allele scratch lives in the codes
    ws.code_cells = (static_cast<size_t>(P.max_haps) * (2 * max_len + 8) / 2 + 64 + 7) & ~size_t(7);
  ws.row_cells = static_cast<size_t>(ws.row_slots) * 3 * ws.band_stride;
  ws.fcode_cells = (static_cast<size_t>(pn + 2) * (max_len + 16) + 7) & ~size_t(7);
  ws.frow_cells = static_cast<size_t>(ws.row_slots) * 3 * ws.w_stride;
  size_t const full_bytes = ((ws.fcode_cells * 2 + 255) & ~size_t(255)) + ((ws.frow_cells * 4 + 255) & ~size_t(255));

  size_t const img_bytes = ws.split ? ((static_cast<size_t>(ws.img_words) * 4 + 255) & ~size_t(255)) : 0;
  ws.img_words = static_cast<u32>(ws.split ? img_bytes / 4 : ws.img_words);
  // MA_POA_SCHED: 1 (default) = the persistent kernel k_poa schedules the windows on the device; 0 = rounds 3-5's host-counted
  // rounds of k_msa / k_msa_band (same device code: msa_window, band_job; same results, tested)
  bool const sched = ws.split && ws.lean && !(getenv("MA_POA_SCHED") && atoi(getenv("MA_POA_SCHED")) == 0);
  ws.full_by_worker = sched ? 1u : 0u;
  // workgroups of the persistent kernel: what the chip holds at once (two per CU while the graph fits 80 KB of LDS; a lane
  // that runs beside others may be told to take one -- MA_POA_WGS_PER_CU -- and leave the other half of every CU's LDS to them)
  int n_cu = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
  }
  int wgs_per_cu = lds > 80 * 1024 ? 1 : 2;
  if (const char* e = getenv("MA_POA_WGS_PER_CU")) wgs_per_cu = std::max(1, std::min(wgs_per_cu, atoi(e)));
  size_t const per_window = ws.code_cells * 2 + ws.row_cells * 4 + (static_cast<size_t>(pn) + 8) * 4 + img_bytes + (sched ? 0 : full_bytes);
  (void)rounds;
  // The POA stage works in the ASSEMBLY stage's workspace: a lane runs its stages one after the other, and what the assembly
  // kernels leave there is dead once the haplotypes are in the caller's buffers (ma_asm_out_t).  One arena for both instead of
  // two that are never in use together: the assembly stage plans with twice the share it had (assemble.hip).
  size_t budget = stage_budget(0.30, ctx->ws_build.cap, size_t(24) << 30, ctx->hbm_share);
  if (const char* e = getenv("MA_WS_GB")) budget = static_cast<size_t>(atoi(e)) << 30;
  // the persistent kernel's workgroups (each with a full-fill area of its own): what the chip holds, what the batch can use,
  // and no more than half the stage's memory
  size_t const max_workers = sched ? std::max<size_t>(1, std::min<size_t>(std::min<size_t>(n, static_cast<size_t>(n_cu) * wgs_per_cu),
                                                                        budget / 2 / full_bytes)) : 0;
  size_t const fixed_bytes = max_workers * full_bytes;
  int const chunk = static_cast<int>(std::max<size_t>(1, std::min<size_t>(n, (budget > fixed_bytes ? budget - fixed_bytes : 0) / per_window)));
  u32 qcap = 64;
  while (qcap < static_cast<u32>(chunk)) qcap <<= 1;
  MA_HIP(ctx, ctx->ws_build.reserve(per_window * static_cast<size_t>(chunk) + fixed_bytes + 16384 + 3 * sizeof(u32) * qcap * kPoaDoms +
                                  sizeof(PoaSched) + sizeof(u32) * qcap));
  u32 const xcd_local = (getenv("MA_POA_XCD") && atoi(getenv("MA_POA_XCD")) == 0) ? 0u : 1u;
  if (getenv("MA_VERBOSE"))
    fprintf(stderr, "[microasm] msa: %d windows, %.2f MB/window + %zu full-fill areas of %.2f MB, budget %.1f GB -> chunks of %d (pn %u, max_len %u, lds %zu)\n",
            n, per_window / 1048576.0, max_workers, full_bytes / 1048576.0, budget / 1073741824.0, chunk, pn, max_len, lds);
  auto kern = lab32 ? (max_len <= 1024 ? k_msa<4, true> : (max_len <= 2048 ? k_msa<8, true> : k_msa<16, true>))
                    : (max_len <= 1024 ? k_msa<4, false> : (max_len <= 2048 ? k_msa<8, false> : k_msa<16, false>));
  MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds)));
  auto pkern = lab32 ? (max_len <= 1024 ? k_poa<4, true> : (max_len <= 2048 ? k_poa<8, true> : k_poa<16, true>))
                     : (max_len <= 1024 ? k_poa<4, false> : (max_len <= 2048 ? k_poa<8, false> : k_poa<16, false>));
  if (sched)
    MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(pkern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
  for (int win0 = 0; win0 < n; win0 += chunk) {
    int const nwin = std::min(chunk, n - win0);
    char* base = static_cast<char*>(ctx->ws_build.p);
    size_t const csz = (static_cast<size_t>(nwin) * ws.code_cells * 2 + 255) & ~size_t(255);
    size_t const rsz = (static_cast<size_t>(nwin) * ws.row_cells * 4 + 255) & ~size_t(255);
    ws.codes = reinterpret_cast<u16*>(base);
    ws.rows = reinterpret_cast<i32*>(base + csz);
    ws.hlast = reinterpret_cast<i32*>(base + csz + rsz);
    size_t const hsz = (static_cast<size_t>(nwin) * (pn + 8) * 4 + 255) & ~size_t(255);
    ws.img = reinterpret_cast<u8*>(base + csz + rsz + hsz);
    // the full fill's areas: one per workgroup of the persistent kernel, one per window for the host-counted rounds
    size_t const n_full = sched ? max_workers : static_cast<size_t>(nwin);
    size_t const isz = (static_cast<size_t>(nwin) * img_bytes + 255) & ~size_t(255);
    size_t const fcsz = (n_full * ws.fcode_cells * 2 + 255) & ~size_t(255);
    size_t const frsz = (n_full * ws.frow_cells * 4 + 255) & ~size_t(255);
    ws.fcodes = reinterpret_cast<u16*>(base + csz + rsz + hsz + isz);
    ws.frows = reinterpret_cast<i32*>(base + csz + rsz + hsz + isz + fcsz);
    {  // counters behind the areas: [0] pending windows of the current launch, [8..15] tier statistics
      u32* ctr = reinterpret_cast<u32*>(base + csz + rsz + hsz + isz + fcsz + frsz);
      ctr = reinterpret_cast<u32*>((reinterpret_cast<uintptr_t>(ctr) + 255) & ~uintptr_t(255));
      ws.pending_ctr = ctr;
      ws.tier_stats = verbose ? ctr + 8 : nullptr;
      MA_HIP(ctx, hipMemsetAsync(ctr, 0, 64, ctx->stream));
    }
    MsaArgs args{b, a, o, ws, P, win0, 0u, 0u};
    if (sched) {
      char* qb = reinterpret_cast<char*>(ws.pending_ctr) + 256;
      PoaSched* const S = reinterpret_cast<PoaSched*>(qb);
      u32* const fq = reinterpret_cast<u32*>(qb + sizeof(PoaSched));
      u32* const gq = fq + static_cast<size_t>(qcap) * kPoaDoms;
      u32* const hq = gq + static_cast<size_t>(qcap) * kPoaDoms;
      u32 const grid = static_cast<u32>(std::min<size_t>(static_cast<size_t>(nwin), max_workers));
      MA_HIP(ctx, hipMemsetAsync(S, 0, sizeof(PoaSched), ctx->stream));
      MA_HIP(ctx, hipMemsetAsync(fq, 0xFF, 3 * sizeof(u32) * qcap * kPoaDoms, ctx->stream));
      u32 const min_fills = static_cast<u32>(getenv("MA_POA_MIN_FILLS") ? atoi(getenv("MA_POA_MIN_FILLS")) : 16);
      static bool const by_work = !(getenv("MA_POA_ORDER") && atoi(getenv("MA_POA_ORDER")) == 0);  // (0: window order, as before)
      u32* const order = by_work ? hq + static_cast<size_t>(qcap) * kPoaDoms : nullptr;
      if (order) hipLaunchKernelGGL(k_poa_order, dim3(1), dim3(1024), 0, ctx->stream, args, order, static_cast<u32>(nwin));
      ctx->tic("k_poa");
      hipLaunchKernelGGL(pkern, dim3(grid), dim3(kT), lds, ctx->stream, args, S, fq, gq, hq, qcap, static_cast<u32>(nwin), xcd_local, min_fills,
                         static_cast<const u32*>(order));
      ctx->toc();
      if (verbose) {
        PoaSched hs;
        MA_HIP(ctx, hipMemcpyAsync(&hs, S, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
        MA_HIP(ctx, ma_stream_sync(ctx));
        fprintf(stderr, "[microasm] k_poa: %u workgroups, %d windows; per XCD (workgroups / windows started):", grid, nwin);
        for (u32 x = 0; x < kPoaDoms; ++x)
          if (hs.dom[x].seen) fprintf(stderr, " %u: %u / %u", x, hs.dom[x].seen, hs.dom[x].started);
        fprintf(stderr, "\n");
      }
    } else if (!ws.split) {
      args.finish = 1;
      ctx->tic("k_msa");
      hipLaunchKernelGGL(kern, dim3(nwin), dim3(kT), lds, ctx->stream, args);
      ctx->toc();
    } else {
      // Per-window progress: k_msa runs every window up to its next banded fill (or to its end); while many windows wait
      // for a fill, k_msa_band fills them (one launch per tier that can hold work) and k_msa resumes them; once few are
      // left, a last k_msa launch finishes them in-kernel (their remaining fills on its wave 0) -- the tail of a batch
      // (third / fourth alignments, retried tiers) no longer costs two launches and a fill's latency per round.
      size_t const band_lds = kStBytes + 16 + 256;  // state block + poa_fill_lean's code table
      u32 const max_rounds = 2u * rounds + 2u;  // every alignment may need two band tiers
      for (u32 r = 0;; ++r) {
        args.round = r;
        args.finish = 0;
        ctx->tic("k_msa");
        hipLaunchKernelGGL(kern, dim3(nwin), dim3(kT), lds, ctx->stream, args);
        ctx->toc();
        u32 pend2[2] = {0, 0};
        MA_HIP(ctx, hipMemcpyAsync(pend2, ws.pending_ctr, 8, hipMemcpyDeviceToHost, ctx->stream));
        MA_HIP(ctx, hipMemsetAsync(ws.pending_ctr, 0, 8, ctx->stream));
        MA_HIP(ctx, ma_stream_sync(ctx));
        u32 const pending = pend2[0], pending_retry = pend2[1];
        if (pending == 0) break;
        // Few windows left: finish them inside k_msa -- unless some of them wait at a RETRIED tier (their 128-column fill
        // failed its certificate: long indels, e.g. the 30-80 base duplications that only assemble further up the k ladder).
        // In-kernel those cost a 256-column fill on one wavefront, or the row-synchronous full fill, per window in turn
        // (12 ms for the slowest window of the ladder workload); a band round costs ~1 ms however few windows it holds.
        if ((pending < min_pending && (pending_retry == 0 || no_retry_rounds)) || r + 1 >= max_rounds) {
          args.round = r + 1;
          args.finish = 1;
          ctx->tic("k_msa");
          hipLaunchKernelGGL(kern, dim3(nwin), dim3(kT), lds, ctx->stream, args);
          ctx->toc();
          break;
        }
        ctx->tic("k_msa_band");
        auto band = [&](auto kfn) { hipLaunchKernelGGL(kfn, dim3(nwin), dim3(64), band_lds, ctx->stream, args); };
        if (ws.lean) band(k_msa_band<true>);  // (first tiers, retried fills and haplotypes that start wide alike)
        else band(k_msa_band<false>);
        ctx->toc();
      }
    }
    if (verbose) {
      u32 tsx[8];
      MA_HIP(ctx, hipMemcpyAsync(tsx, ws.tier_stats, 32, hipMemcpyDeviceToHost, ctx->stream));
      MA_HIP(ctx, ma_stream_sync(ctx));
      fprintf(stderr, "[microasm] msa tiers: fills 64/128/256 = %u/%u/%u, failed certificates = %u/%u/%u\n", tsx[0], tsx[1],
              tsx[2], tsx[4], tsx[5], tsx[6]);
    }
    MA_HIP(ctx, hipGetLastError());
  }
  return MA_OK;
}

}  // namespace ma
