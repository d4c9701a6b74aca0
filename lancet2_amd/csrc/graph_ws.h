// Device workspace of the cbdg assembly stage (build.hip + clean.hip).
//
// HBM layout per active window slot `a` (windows of a chunk that attempt the current k):
//   k-mer table   : tbl_key[a][TC] u64 (0 = empty), tbl_first[a][TC] u32 (first instance),
//                   tbl_cnt[a][TC][S+2] u32 (per-sample + per-role read support)
//   instances     : inst_slot[a][IS] u32 = slot | sign<<30 | errfree<<31, sequence-major
//   mate-mer set  : mm_key[a][MC] u64, mm_min[a][MC] u32        (graph.h:102-117)
//   compact graph : node arrays [a][NC] in canonical (first-insertion) order, see below
//   BFS arena     : arena[a][AC] 16 B records                    (max_flow.h:51-63)
#pragma once
#include "ma_internal.h"

namespace ma {

constexpr int kEdgeCap = 16;       // edges per node (reference: InlinedVector<Edge, 8>, node.h:42)
constexpr u32 kNoNode = 0xFFFFFFFFu;
// instance word: [31] error-free  [30] canonical == as-seen (PLUS)  [29] FAST: k-mer equals the reference
// k-mer at the hinted offset, low bits hold that reference POSITION instead of a table slot
// [28] GEN: read support goes through the general mate-mer set  [27] LAST k-mer of its sequence
// [26:0] slot / reference position
constexpr u32 kInstErrFree = 1u << 31;
constexpr u32 kInstPlus = 1u << 30;
constexpr u32 kInstFast = 1u << 29;
constexpr u32 kInstGen = 1u << 28;
constexpr u32 kInstLast = 1u << 27;
constexpr u32 kInstFirst = 1u << 26;  // set by k_rank: first instance of a node that survives low-coverage pruning
constexpr u32 kInstSlotMask = (1u << 26) - 1;
constexpr int kMaxSamples = 8;

struct GraphWs {
  // ---- per k attempt ----
  int k;                  // the ladder's first k: what a window without a k of its own yet is counted with (capacity planning)
  const u32* win_k;       // [n_windows] the k window w attempts in this pass (0: none yet) -- every window climbs its OWN
                          // ladder, so one pass serves windows at different k (k_select_active)
  const u32* win_kfirst;  // non-null in the nested pass of the speculative ladder tail: window w is built at this k only
  u64 pk1;   // P^(k-1) mod 2^64
  u64 pinv;  // P^-1   mod 2^64
  int n_active;
  const u32* active;      // [n_active] global window ids
  // ---- per batch ----
  u32* seq_inst_base;     // [n_reads + n_windows]: index read_win_off[w] + w is the ref of window w, reads follow
  u32* win_ninst;         // [n_windows] total instances of window w at this k
  u32* win_nread_inst;    // [n_windows] instances that come from reads
  u32 max_reads;          // most reads of any window of the batch
  u32 max_read_len;       // longest read of the batch
  // table
  int tc_log2, mc_log2;   // ALLOCATED strides (log2): the table stride an attempt uses is tbl_log2(ws), sized on the device
  const u32* max_slow;    // [1] device: most slow-path instances of any window of this attempt (k_classify) -- what sizes the
                          //     table stride; read by the kernels themselves, so the host never waits for it
  u32 mm_probe_max;  // k_mm_lds: probes after which its LDS set counts as full (tests lower it: MA_MM_PROBE_MAX)
  u32 mm_force_hbm;  // capacity retry passes: every window's mate-mers through the HBM set (k_mm_insert / k_count)
  u32 inst_stride;
  u64* tbl_key;
  u32* tbl_first;
  u32* tbl_cnt;
  u32* inst_slot;
  u32* slowq;             // [a][inst_stride] (seq << 12 | offset) of the instances that need the hash table
  u32* n_slow;            // [a]
  u32* mm_mode;           // [a] general mate-mer instances of the window; bit31: needs the HBM-resident set
  u32* win_nslots;        // [a] table slots this window uses (<= the stride; 6144 = k_insert's LDS map copied out, else a power of two)
  uint4* slow_rec;        // [a][inst_stride] k_insert: one record per reference k-mer, then per slow-queue entry: id, instance | flags, sequence
  u32 graph_fused;        // k_graph runs in this attempt (then k_support may queue a window's read-support counts for it)
  u32 dd_log2;            // log2 of the entries of a wavefront's dedup table in k_support
  u32* n_genq;            // [a] entries of the window's key / count queue (k_support; the queue lives where slowq did)
  u32* n_edgeq;           // [a] entries of the window's edge queue (k_support; the queue lives where slow_rec did)
  u32* gr_done;           // [a] 1: k_graph produced the window's node records and edges (k_rank / k_edges / k_edge_sort skip it)
  u8* rd_flag;            // [n_reads] general-path k-mer of this read hit a reference node
  // HBM-resident mate-mer sets (windows without mapping hints, names in separate runs, capacity retries): carved out of a
  // pool by the windows that need one (k_support: one atomicAdd), so that nothing is reserved per window on the usual route and
  // the host never has to learn how many windows need how much
  u8* mm_pool;
  unsigned long long mm_pool_bytes;
  unsigned long long* mm_pool_used;  // [1] device bump pointer
  unsigned long long* mm_off;        // [a] the window's set: byte offset into mm_pool (keys u64[cap], then minima u32[cap])
  u32* mm_log2;                      // [a] log2 of its capacity
  // compact graph (per active slot, NC nodes)
  u32 nc;                 // node capacity per window
  u32* slot_node;         // [a][TC] slot -> node idx (kNoNode if pruned) 
  u32* n_nodes;           // [a]
  u32* nd_cnt;            // [a][NC][S]
  u32* nd_role;           // [a][NC][2]
  u32* nd_src;            // [a][NC] source of the canonical k-mer: bit31 = read buffer (else ref), low bits = byte
                          //          offset relative to the window's first ref / read byte
  u8* nd_label;           // [a][NC]
  u8* nd_sign;            // [a][NC] 1 = PLUS
  u8* nd_nedge;           // [a][NC]
  u32* nd_edge;           // [a][NC][kEdgeCap]  dst << 2 | kind
  u32* nd_ekey;           // [a][NC][kEdgeCap]  insertion-order key (build only)
  u32* ref_node;          // [a][max_ref_kmers] node idx of each reference k-mer (kNoNode if pruned): mRefNodeIds
  u32* ref_slot;          // [a][max_ref_kmers] table slot of each reference k-mer
  u32 ref_stride;
  u32 max_ref_len;
  // clean-stage scratch
  u32* nd_comp;           // [a][NC]
  u32* nd_len;            // [a][NC]
  u8* nd_alive;           // [a][NC]
  u32* nd_head;           // [a][NC] slice list head / tail (original node ids)
  u32* nd_tail;
  u32* sl_next;           // [a][NC]
  u32* sl_prev;
  u32* sl_desc;           // [a][NC] start | len << 8 | rc << 16
  u32* scratch;           // [a][4 * NC] queues / flat indices / sort buffers
  u32 ac;                 // arena capacity (records)
  uint4* arena;           // [a][AC]
  u32* win_flags;         // [n_windows] bit0 done, bit1 retry-at-next-k requested
  int num_samples;
  // ---- compact graph: what the first CompressGraph leaves of the candidate components (k_clean_chains -> k_clean_tail) ----
  u32 vc;                 // node capacity per window (all candidate components together)
  u32 cg_sc;              // node stride of the tail's scratch (>= vc: walk pool and search tables scale with it)
  u32 pool_cap;           // bytes of merged node strings per window
  u32* cg_state;          // [a] 0: not produced -- k_clean takes the window from the raw graph; 1: produced
  u32* cg_hdr;            // [a][kCgHdr] n, ncand, then per candidate: comp, size, src, snk, soff, koff (compact ids)
  u32* cg_cnt;            // [a][vc][S]
  u32* cg_role;           // [a][vc][2]
  u32* cg_bsrc;           // [a][vc] base string of the node: nd_src of the original k-mer, or bit30 | offset into cg_pool
  u32* cg_blen;           // [a][vc] its length
  u32* cg_len;            // [a][vc]
  u32* cg_comp;           // [a][vc] candidate's component id
  u8* cg_label;           // [a][vc]
  u8* cg_sign;            // [a][vc]
  u8* cg_bsign;           // [a][vc]
  u8* cg_nedge;           // [a][vc]
  u8* cg_alive;           // [a][vc]
  u32* cg_edge;           // [a][vc][kCgEdgeCap]
  u32* cg_head;           // [a][vc] slice lists of the tail (slice id = compact node id)
  u32* cg_tail;
  u32* cg_snext;
  u32* cg_sprev;
  u32* cg_sdesc;
  u32* cg_scratch;        // [a][32 * cg_sc]
  u8* cg_pool;            // [a][pool_cap]
};
constexpr int kCgEdgeCap = 8;
constexpr int kCgMaxCand = 15;
constexpr int kCgHdr = 8 + 6 * 16;

// the k window w is built / cleaned with in the current pass
__device__ __forceinline__ int win_kmer(GraphWs const& ws, int w) {
  u32 const k = ws.win_k[w];
  return k ? static_cast<int>(k) : ws.k;
}


// table stride of this attempt (log2): the host's old formula -- slots for the busiest window's slow instances + reference
// k-mers at a load of 3/4, at least 2^10, at most what was allocated -- evaluated where max_slow lives
__device__ __forceinline__ int tbl_log2(GraphWs const& ws) {
  u32 const need = (ws.max_slow[0] + ws.ref_stride) / 3u * 4u + 32u;
  int tc = 32 - __clz(static_cast<int>(need - 1u));
  tc = tc < 10 ? 10 : tc;
  return tc < ws.tc_log2 ? tc : ws.tc_log2;
}

__device__ __forceinline__ u64 dev_fmix64(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

}  // namespace ma
