// Colored de Bruijn graph construction on gfx950 -- replaces Graph::BuildGraph / AddNodes
// (cbdg/graph.cpp:262-341), the MateMer read-support dedup (cbdg/graph.h:102-117), the first
// RemoveLowCovNodes(0) pass (graph.cpp:135, :363-390) and produces the compact, canonically ordered
// node/edge arrays the cleaning kernel (clean.hip) walks.
//
// One workgroup per window attempt and kernel.  Data-parallel passes over the k-mer INSTANCES of the window
// (reference k-mers first, then each filter-passing read in collector order -- that sequence-major order is
// the canonical first-insertion order of DESIGN.md); every instance owns one 32-bit instance word (graph_ws.h):
//   k_classify   wave per tile of 64 reads staged in LDS: error-free bit from sequential f64 Phred prefix sums
//                (graph.cpp:280-304, two lagged accumulators); k-mers equal to the reference k-mer at the read's
//                hinted offset become FAST instances (no hashing), the others go to the window's slow queue
//   k_insert     distinct slow + reference k-mers in an LDS map (canonical decision kmer.cpp:17-28, polynomial +
//                fmix64 id), then one open-addressing insert per distinct k-mer; first instance = try_emplace
//                (graph.cpp:325-326)
//   k_support    read support of FAST instances with the mate-mer rule (graph.h:102-117), wave per group of mates
//   k_mm_lds     (qname, role, node) set of the remaining instances in LDS + their support (node.cpp:18-24);
//                k_mm_insert / k_count: the HBM-resident fallback of that set
//   k_rank       low-coverage pruning + canonical ranking of survivors, node records
//   k_edges      forward + mirror edge of every (k+1)-mer whose two nodes survive, distinct edges collected in an
//                LDS set with the order key of the first occurrence; k_edge_sort orders them.
#include "graph_ws.h"

namespace ma {

constexpr int kBT = 256;
__constant__ u64 c_phred_bits[256] = {
#include "../../include/ma_phred_lut.inc"
};

struct SeqInfo {
  u64 off;   // byte offset in its buffer
  u32 len;
  u32 nk;    // number of k-mer instances (0 if len < k+1: SlidingView(seq, k+1) is empty)
};

__device__ __forceinline__ u32 seq_count(const DBatch& b, int w) {
  return 1u + (b.read_win_off[w + 1] - b.read_win_off[w]);
}

// sequence s of window w: s == 0 is the reference, s >= 1 is read (read_win_off[w] + s - 1)
__device__ __forceinline__ SeqInfo seq_info(const DBatch& b, int w, u32 s, int k) {
  SeqInfo si;
  if (s == 0) {
    si.off = b.ref_off[w];
    si.len = b.ref_off[w + 1] - b.ref_off[w];
  } else {
    u32 const r = b.read_win_off[w] + s - 1;
    si.off = b.read_off[r];
    si.len = static_cast<u32>(b.read_off[r + 1] - b.read_off[r]);
    if (!(b.read_flags[r] & MA_RF_PASS)) si.len = 0;  // graph.cpp:275
  }
  si.nk = si.len >= static_cast<u32>(k) + 1 ? si.len - k + 1 : 0;
  return si;
}

// block-wide exclusive scan helper over a small per-thread value: wave scans by shuffles, wave totals through LDS
// (two barriers; sh holds at least kBT / 64 words)
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* sh, u32* total) {
  int const t = threadIdx.x, lane = t & 63, wave = t >> 6;
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 const y = __shfl_up(inc, d);
    if (lane >= d) inc += y;
  }
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  u32 before = 0, tot = 0;
#pragma unroll
  for (int x = 0; x < kBT / 64; ++x) {
    u32 const c = sh[x];
    before += x < wave ? c : 0u;
    tot += c;
  }
  *total = tot;
  __syncthreads();
  return before + inc - v;
}

// ---- per-k instance bookkeeping: seq_inst_base + totals, over ALL windows of the chunk ----
__global__ __launch_bounds__(kBT) void k_count_inst(DBatch b, GraphWs ws, int win0, int nwin, u32* maxima) {
  __shared__ u32 sh[kBT];
  int const w = win0 + blockIdx.x;
  if (blockIdx.x >= static_cast<u32>(nwin)) return;
  if (ws.win_flags[w] & 1u) return;  // resolved in an earlier pass: no instances to plan for
  int const kw = win_kmer(ws, w);
  u32 const ns = seq_count(b, w);
  u32 const base_idx = b.read_win_off[w] + w;
  u32 running = 0, read_inst = 0, max_len = 0;
  for (u32 s0 = 0; s0 < ns; s0 += kBT) {
    u32 const s = s0 + threadIdx.x;
    u32 nk = 0;
    if (s < ns) {
      SeqInfo const si = seq_info(b, w, s, kw);
      nk = si.nk;
      if (s > 0) max_len = max(max_len, si.len);
    }
    u32 tot;
    u32 const ex = block_excl_scan(nk, sh, &tot);
    if (s < ns) ws.seq_inst_base[base_idx + s] = running + ex;
    if (s0 == 0) {
      // the reference contributes sh[0]-lane value; recompute below
    }
    running += tot;
  }
  if (threadIdx.x == 0) {
    u32 const refk = seq_info(b, w, 0, kw).nk;
    read_inst = running - refk;
    ws.win_ninst[w] = running;
    ws.win_nread_inst[w] = read_inst;
    atomic_max_lazy(&maxima[0], running);
    atomic_max_lazy(&maxima[1], read_inst);
    atomic_max_lazy(&maxima[2], refk);
    atomic_max_lazy(&maxima[3], ns - 1);
  }
  // (one report per wavefront, not one per thread: every thread of every window used to send its longest read)
  for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, static_cast<u32>(__shfl_xor(max_len, off)));
  if ((threadIdx.x & 63) == 0 && max_len) atomic_max_lazy(&maxima[4], max_len);
}

// ---- choose the windows of this pass and the k each of them attempts (graph.cpp:106-120) ----
// The reference climbs the ladder min_k, min_k + step, ... per window: a k whose repeat gate fires is skipped, a k whose
// graph has a cycle / is too complex sends the window to the next one.  Windows are independent, so every pending window
// takes ITS next k here and one pass of the build / clean kernels serves all of them, each at its own k (a pass per k
// of the ladder cost twenty passes of mostly idle kernels).  win_k[w] = Graph::CurrentK(): the k attempted last; a window
// that runs out of ladder ends on its last rung, unresolved, as the reference's loop does.
__global__ void k_select_active(GraphWs ws, int win0, int nwin, const u32* gate_approx, u32* win_k, u32* active,
                                u32* n_active, int min_k, int max_k, int k_step) {
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nwin) return;
  int const w = win0 + i;
  if (ws.win_flags[w] & 1u) return;           // done in an earlier pass
  u32 const last = win_k[w];
  if (ws.win_kfirst) {  // nested pass of the speculative ladder tail: one rung, chosen by the caller (gate already applied)
    if (last) {
      atomicOr(&ws.win_flags[w], 1u);         // its one attempt is over
      return;
    }
    atomicAdd(n_active + 1, 1u);
    win_k[w] = ws.win_kfirst[w];
    active[atomicAdd(n_active, 1u)] = static_cast<u32>(w);
    return;
  }
  u32 k = last ? last + static_cast<u32>(k_step) : static_cast<u32>(min_k);
  u32 const ga = gate_approx[w];
  if (k <= ga) k += ((ga - k) / static_cast<u32>(k_step) + 1u) * static_cast<u32>(k_step);  // HasExactOrApproxRepeat -> continue
  if (k > static_cast<u32>(max_k)) {          // ladder exhausted
    win_k[w] = static_cast<u32>(min_k + (max_k - min_k) / k_step * k_step);
    atomicOr(&ws.win_flags[w], 1u);
    return;
  }
  atomicAdd(n_active + 1, 1u);                // still pending
  win_k[w] = k;                                // Graph::CurrentK()
  u32 const a = atomicAdd(n_active, 1u);
  active[a] = static_cast<u32>(w);
}

// instance index -> sequence: the largest s with base[s] <= ii (sequences without k-mers share the base of
// their successor, so the last one of an equal run is the one that owns the instance)
constexpr u32 kSeqCap = 2048;  // sequences whose instance bases are cached in LDS
__device__ __forceinline__ u32 seq_of(const u32* base, u32 ns, u32 ii) {
  u32 lo = 0, hi = ns;
  while (hi - lo > 1) {
    u32 const mid = (lo + hi) >> 1;
    if (base[mid] <= ii) lo = mid; else hi = mid;
  }
  return lo;
}
// stage the window's per-sequence instance bases in LDS when they fit (callers __syncthreads afterwards)
__device__ __forceinline__ const u32* stage_seq_bases(const u32* gbase, u32 ns, u32* l_base) {
  if (ns > kSeqCap) return gbase;
  for (u32 s = threadIdx.x; s < ns; s += blockDim.x) l_base[s] = gbase[s];
  return l_base;
}

// (flags: the window's flag word -- once the table has been found full (bit 2) nobody probes all of it again: a first pass that
//  plans a quarter of a deep window's instances makes "full" an expected event, and every id without room walked 2^18 slots)
__device__ __forceinline__ u32 table_insert(u64* keys, u32 mask, u64 id, const u32* flags) {
  u32 slot = static_cast<u32>(id) & mask;
  for (u32 probe = 0; probe <= mask; ++probe) {
    if ((probe & 255u) == 255u && (*reinterpret_cast<const volatile u32*>(flags) & 4u)) return kNoNode;
    u64 cur = keys[slot];
    if (cur == id) return slot;
    if (cur == 0) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[slot]), 0ull,
                                         static_cast<unsigned long long>(id));
      if (old == 0ull || old == id) return slot;
    }
    slot = (slot + 1) & mask;
  }
  return kNoNode;
}

__device__ __forceinline__ u32 table_find(const u64* keys, u32 mask, u64 id) {
  u32 slot = static_cast<u32>(id) & mask;
  for (u32 probe = 0; probe <= mask; ++probe) {
    u64 cur = keys[slot];
    if (cur == id) return slot;
    if (cur == 0) return kNoNode;
    slot = (slot + 1) & mask;
  }
  return kNoNode;
}

// canonical decision of cbdg/kmer.cpp:17-28 on bytes s[0..k)
__device__ __forceinline__ bool canon_plus(const u8* s, int k) {
  int const half = (k + 1) / 2;
  for (int i = 0; i < half; ++i) {
    signed char const f = static_cast<signed char>(s[i]);
    signed char const r = static_cast<signed char>(dev_complement(s[k - 1 - i]));
    if (f < r) return true;
    if (f > r) return false;
  }
  return true;
}

// ---- hinted build -------------------------------------------------------------------------------------
// Reads arrive pre-aligned (they come from a BAM): ~85-90 % of their k-mers are byte-identical to the
// reference k-mer at the read's aligned offset.  Such an instance IS the reference node at that position
// (same string => same canonical k-mer), so it needs no hashing, no table probe and no HBM traffic; its
// read support is accumulated in LDS counters per reference position.  Only k-mers that overlap a
// mismatch / indel / clipped base (or reads without a hint) take the general hash-table path.
// Exactness: the hint only selects the path; a wrong hint merely sends the k-mer down the general path.
//
// Mate-mer de-duplication (graph.h:102-117) for the fast path: reads with the same (qname, role, sample)
// are adjacent in collector order; one thread owns such a GROUP and keeps a bitmask of the offsets of
// the first member that were counted, so the second member's k-mer at the same reference position is
// recognised as a duplicate.  Groups where this local reasoning could be wrong (more than two members,
// reads longer than the mask, a general-path k-mer that turns out to be a reference node, a qname shared
// by two samples of one role) are routed wholesale through the general mate-mer set (GEN bit).
constexpr int kMaskWords = 10;  // fast-path bitmask covers reads with <= 320 k-mers
constexpr u32 kMmLdsCap = 32768;              // entries of the LDS mate-mer set of k_mm_lds
constexpr u32 kMmLdsMax = kMmLdsCap * 7 / 8;  // general instances per window it accepts (distinct keys are fewer)
constexpr u32 kMmRunMax = 32;  // longest run of one (qname, role) in a window that k_mm_lds takes in chunks

struct BuildLds {
  u8* ref;        // [max_ref_len + 8]
  u32* ref_slot;  // [ref_stride]   slot | plus << 30
  u32* cnt;       // [ref_stride * (S + 2)]
  u32* mask;      // [kMaskWords][kBT]
  u32* xs_key;    // [kXs]  (qname << 1 | role) + 1
  u8* xs_sample;  // [kXs]
};
constexpr u32 kXs = 1024;

__device__ __forceinline__ u32 lds_bytes_build(u32 max_ref_len, u32 ref_stride, int S) {
  return ((max_ref_len + 8 + 15) & ~15u) + 4u * ref_stride + 4u * ref_stride * (S + 2) + 4u * kMaskWords * kBT + 4u * kXs +
         kXs;
}

// Shared prologue: stage the window's reference bytes in LDS.
__device__ __forceinline__ void stage_ref(const DBatch& b, const SeqInfo& rsi, u8* l_ref, u32 cap) {
  const u8* s = b.ref_bases + rsi.off;
  for (u32 i = threadIdx.x; i < rsi.len && i < cap; i += kBT) l_ref[i] = s[i];
}
__device__ __forceinline__ bool same_group(const DBatch& b, u32 ra, u32 rb) {
  return (b.read_flags[rb] & MA_RF_PASS) && b.read_qname_id[ra] == b.read_qname_id[rb] &&
         ((b.read_flags[ra] ^ b.read_flags[rb]) & MA_RF_CASE) == 0 && b.read_sample[ra] == b.read_sample[rb];
}

// (1) k_classify: every read instance is classified WITHOUT touching the hash table.  One wavefront per tile
//     of 64 consecutive reads (one read per lane: the f64 Phred prefix sums are a serial chain per read).
//     The tile's bases and qualities are contiguous in the batch, so they are staged in LDS with coalesced
//     loads -- a lane walking its read byte by byte straight from HBM costs one cache-line fetch per byte
//     once thousands of lanes do it (measured: ~180x read amplification).  The lane's loop takes FOUR k-mer
//     positions per trip with every LDS load of the trip unconditional (indices clamped instead of guarded): the 24
//     byte loads go out together, then the 8 Phred look-ups, and only the sums and the mismatch count run as a
//     chain.  (One position per trip with guarded loads was 16 dependent LDS round trips per four positions: the
//     compiler cannot hoist a load out of a branch.)  Instances that need the general path are remembered in a
//     per-lane bitmask and appended to the window's slow queue at the end with one atomic per wave, so that
//     k_insert can hash them with full lanes.
#ifdef MA_PROFILE
__device__ unsigned long long g_iprof[32];
#define IPROF_T0() unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define IPROF(slot)                                                            \
  do {                                                                         \
    __syncthreads();                                                           \
    unsigned long long _t1 = __builtin_amdgcn_s_memtime();                     \
    if (threadIdx.x == 0) atomicAdd(&g_iprof[slot], _t1 - _t0);                \
    _t0 = _t1;                                                                 \
  } while (0)
#else
#define IPROF_T0() do {} while (0)
#define IPROF(slot) do {} while (0)
#endif
// a barrier that is also a phase mark of the profile build
#define IPROF_SYNC(slot) do { __syncthreads(); IPROF(slot); } while (0)
#ifdef MA_PROFILE
#define KPROF(slot)                                                            \
  do {                                                                         \
    unsigned long long _t1 = __builtin_amdgcn_s_memtime();                     \
    if (threadIdx.x == 0) atomicAdd(&g_iprof[slot], _t1 - _t0);                \
    _t0 = _t1;                                                                 \
  } while (0)
#else
#define KPROF(slot) do {} while (0)
#endif
constexpr u32 classify_mask_words(u32 max_read_len) { return (max_read_len + 31) / 32 + 1; }  // per read: slow-instance bits
// slow-queue item: [11:0] offset of the k-mer in its sequence, [28:12] sequence, [29] last k-mer of its sequence, [30] error-free,
// [31] canonical == as-seen (set by k_insert once it has hashed the k-mer) -- everything k_insert needs to write the instance
// word without reading it first
constexpr u32 kQPlus = 1u << 31, kQErrFree = 1u << 30, kQLast = 1u << 29, kQSeqMask = 0x1FFFFu;
// Round 6: TWO wavefronts per tile.  The tile's 22 KB of staged reads allowed 1.75 wavefronts per SIMD, each a serial chain
// per read (~170 vector instructions per trip of four positions): 0.15 instructions per SIMD and cycle.  Both wavefronts map a
// lane to a read; wavefront 0 takes the k-mer positions below `split` (half of the longest read's, a multiple of 32: the
// mask words of the two halves are disjoint), wavefront 1 the rest -- after a pre-pass that adds the read's Phred values up
// to `split` + k in the reference's order (prefix sums are sequential by definition: the same additions, the same bits) and
// counts the mismatches of the window it starts at.  Twice the wavefronts on the same LDS, 65 % of the chain.
constexpr u32 kClsT = 128;
__global__ __launch_bounds__(kClsT) void k_classify(DBatch b, GraphWs ws, u32* max_slow, u32 tiles_per_win, u32 tile_cap) {
  extern __shared__ unsigned char lds_build[];
  int const a = blockIdx.x / tiles_per_win;
  u32 const tile = blockIdx.x % tiles_per_win;
  int const w = static_cast<int>(ws.active[a]);
  int const k = win_kmer(ws, w);
  int const lane = threadIdx.x & 63;
  u32 const half = threadIdx.x >> 6;  // which wavefront of the tile
  u32 const ns = seq_count(b, w);
  u32 const nreads = ns - 1;
  if (tile * 64 >= nreads) return;
  IPROF_T0();
  u32 const cnt = min(64u, nreads - tile * 64);
  u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  u32* slowq = ws.slowq + static_cast<size_t>(a) * ws.inst_stride;
  u32 const base_idx = b.read_win_off[w] + w;
  // LDS carve
  u32 const ref_cap = (ws.max_ref_len + 8 + 15) & ~15u;
  u32 const MW = classify_mask_words(ws.max_read_len);
  u8* l_ref = lds_build;
  f64* l_phred = reinterpret_cast<f64*>(lds_build + ref_cap);
  u8* l_bases = lds_build + ref_cap + 2048;
  u8* l_quals = l_bases + tile_cap;
  u32* l_mask = reinterpret_cast<u32*>(l_quals + tile_cap);  // [64][MW] bit o of read (lane): instance o is slow
  u32* l_emask = l_mask + 64u * MW;                           // [64][MW] bit o: instance o is error-free
  SeqInfo const rsi = seq_info(b, w, 0, k);
  i32 const ref_len = static_cast<i32>(rsi.len);
  {
    const u8* s = b.ref_bases + rsi.off;
    for (u32 i = threadIdx.x; i < rsi.len && i < ws.max_ref_len + 8; i += kClsT) l_ref[i] = s[i];
  }
  for (u32 i = threadIdx.x; i < 256; i += kClsT) reinterpret_cast<u64*>(l_phred)[i] = c_phred_bits[i];
  u32 const r0 = b.read_win_off[w] + tile * 64;
  u64 const byte0 = b.read_off[r0], byte1 = b.read_off[r0 + cnt];
  u64 const al0 = byte0 & ~static_cast<u64>(15);
  u64 const total_end = b.read_off[b.n_reads];
  {
    // coalesced copy in 16-byte words, four per lane and array in flight (one 4-byte word per trip made this copy 36 %
    // of the kernel, four of them still a third: ~19 dependent round trips to HBM); the last word is read byte-wise if
    // it would cross the end of the batch
    u64 const nwords = (byte1 - al0 + 15) >> 4;
    constexpr int kSU = 4;
    for (u64 x0 = threadIdx.x; x0 < nwords; x0 += kClsT * kSU) {
      uint4 vb[kSU], vq[kSU];
#pragma unroll
      for (int u = 0; u < kSU; ++u) {
        u64 const x = x0 + kClsT * u, at = al0 + 16 * x;
        vb[u] = vq[u] = make_uint4(0, 0, 0, 0);
        if (x < nwords) {
          if (at + 16 <= total_end) {
            vb[u] = *reinterpret_cast<const uint4*>(b.read_bases + at);
            vq[u] = *reinterpret_cast<const uint4*>(b.read_quals + at);
          } else {
            u32 tb[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
            for (u64 y = 0; at + y < total_end; ++y) {
              tb[y >> 2] |= static_cast<u32>(b.read_bases[at + y]) << (8 * (y & 3));
              tq[y >> 2] |= static_cast<u32>(b.read_quals[at + y]) << (8 * (y & 3));
            }
            vb[u] = make_uint4(tb[0], tb[1], tb[2], tb[3]);
            vq[u] = make_uint4(tq[0], tq[1], tq[2], tq[3]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kSU; ++u) {
        u64 const x = x0 + kClsT * u;
        if (x < nwords && 16 * x + 16 <= tile_cap) {
          reinterpret_cast<uint4*>(l_bases)[x] = vb[u];
          reinterpret_cast<uint4*>(l_quals)[x] = vq[u];
        }
      }
    }
  }
  __syncthreads();
  KPROF(8);
  bool const hints = b.read_hint != nullptr && rsi.len <= ws.max_ref_len + 8 && ref_len > 0;
  u32 nslow = 0;
  bool all_slow = false;
  u32 s_idx = 0, my_nk = 0;
  // this wavefront's k-mer positions of every read: [o_begin, o_end)
  u32 const nk_longest = ws.max_read_len >= static_cast<u32>(k) ? ws.max_read_len - static_cast<u32>(k) + 1u : 1u;
  u32 const split = max(32u, (nk_longest / 2u + 16u) & ~31u);
  u32 my_begin = 0, read_nk = 0;
  if (lane < static_cast<int>(cnt)) {
    s_idx = 1 + tile * 64 + lane;
    SeqInfo const si = seq_info(b, w, s_idx, k);
    u32 const r = r0 + lane;
    if (half == 0) ws.rd_flag[r] = 0;
    u64 const rel = si.off - al0;
    bool const fits = rel + si.len <= tile_cap;  // always true: tile_cap covers 64 reads of the longest length
    u32 const o_begin = half ? split : 0u, o_end = half ? si.nk : min(si.nk, split);
    if (si.nk != 0 && fits && o_begin < o_end) {
      my_nk = o_end;
      my_begin = o_begin;
      read_nk = si.nk;
      const u8* s = l_bases + rel;
      const u8* q = l_quals + rel;
      u32 const ibase = ws.seq_inst_base[base_idx + s_idx];
      i32 const hint = hints ? b.read_hint[r] : MA_NO_HINT;
      bool const use_hint = hint != MA_NO_HINT && hint > -100000 && hint < 100000 && si.nk <= 32u * kMaskWords;
      all_slow = !use_hint;
      i32 const h0 = use_hint ? hint : 0;
      u32 const ku = static_cast<u32>(k), nk = si.nk;
      auto in_ref = [&](u32 i) -> bool {
        i32 const rp = h0 + static_cast<i32>(i);
        return rp >= 0 && rp < ref_len;
      };
      f64 lead = 0.0, lag = 0.0;  // prefix[o+k] and prefix[o] of graph.cpp:283-285
      i32 mm = 0;                 // mismatches of read[o, o+k) against ref[hint+o, ...)
      // Round 6: FOUR bytes per LDS load.  The lane loop was bound by the LDS pipeline, not by latency: 24 one-byte loads per
      // trip of four positions, each lane at its own read's offset (150 bytes apart: bank conflicts) -- ~2100 cycles per
      // trip with seven wavefronts on a CU.  A stream of consecutive bytes at ANY offset is one aligned dword per trip and a
      // byte-align with the previous one (v_alignbyte); the reference stream, whose offset is the read's hint and may leave
      // the staged window on either side, takes two clamped dwords per trip (a byte that is in range always comes from an
      // unclamped index; what lies outside is never looked at: in_ref()).  16 LDS loads per trip instead of 32.
      u32 const* const lds32 = reinterpret_cast<u32 const*>(lds_build);
      u32 const q_byte0 = static_cast<u32>(q - lds_build), s_byte0 = static_cast<u32>(s - lds_build);
      i32 const ref_words_m1 = static_cast<i32>(ref_cap >> 2) - 1;
      // bytes [byte, byte + 4) of the staged tile: `prev` = the aligned dword that holds `byte` (kept from the trip before)
      auto next4 = [&](u32 byte, u32& prev) -> u32 {
        u32 const nxt = lds32[(byte >> 2) + 1u];
        u32 const v = __builtin_amdgcn_alignbyte(nxt, prev, byte & 3u);
        prev = nxt;
        return v;
      };
      auto ref4 = [&](i32 pos) -> u32 {  // reference bytes at window positions [pos, pos + 4)
        i32 const wi = pos >> 2;
        u32 const w0 = lds32[static_cast<u32>(min(max(wi, 0), ref_words_m1))], w1 = lds32[static_cast<u32>(min(max(wi + 1, 0), ref_words_m1))];
        return __builtin_amdgcn_alignbyte(w1, w0, static_cast<u32>(pos) & 3u);
      };
      // wavefront 1's pre-pass: prefix[o_begin] by the same sequential additions the reference's prefix array is made of
      // (the loop below then carries on from prefix[o_begin + k] and the mismatch count of the window at o_begin)
      if (o_begin) {
        u32 pq = lds32[q_byte0 >> 2];
        for (u32 i = 0; i < o_begin; i += 4) {  // (o_begin is a multiple of 32)
          u32 const q4 = next4(q_byte0 + i, pq);
          f64 pv[4];
#pragma unroll
          for (u32 j = 0; j < 4; ++j) pv[j] = l_phred[(q4 >> (8 * j)) & 0xFFu];
#pragma unroll
          for (u32 j = 0; j < 4; ++j) lag = (i + j == 0) ? pv[j] : lag + pv[j];
        }
        lead = lag;
      }
      for (u32 i = 0; i < ku; i += 4) {
        u32 qv[4], sv[4], rv[4];
        {
          u32 pq = lds32[(q_byte0 + o_begin + i) >> 2], ps = lds32[(s_byte0 + o_begin + i) >> 2];
          u32 const q4 = next4(q_byte0 + o_begin + i, pq), s4 = next4(s_byte0 + o_begin + i, ps), r4 = ref4(h0 + static_cast<i32>(o_begin + i));
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            qv[j] = (q4 >> (8 * j)) & 0xFFu;
            sv[j] = (s4 >> (8 * j)) & 0xFFu;
            rv[j] = (r4 >> (8 * j)) & 0xFFu;
          }
        }
        f64 pv[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) pv[j] = l_phred[qv[j]];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          if (i + j < ku) {
            lead = (o_begin + i + j == 0) ? pv[j] : lead + pv[j];
            mm += (use_hint && in_ref(o_begin + i + j) && sv[j] == rv[j]) ? 0 : 1;
          }
        }
      }
      u32* mw = l_mask + static_cast<u32>(lane) * MW;
      u32* ew = l_emask + static_cast<u32>(lane) * MW;
      u32 sacc = 0, eacc = 0;
      u32 pqo = lds32[(q_byte0 + o_begin) >> 2], pqi = lds32[(q_byte0 + o_begin + ku) >> 2], pso = lds32[(s_byte0 + o_begin) >> 2],
          psi = lds32[(s_byte0 + o_begin + ku) >> 2];
      for (u32 o = o_begin; o < o_end; o += 4) {
        u32 qo[4], qi[4], so[4], si4[4], ro[4], ri[4];
        {
          u32 const qo4 = next4(q_byte0 + o, pqo), qi4 = next4(q_byte0 + o + ku, pqi);
          u32 const so4 = next4(s_byte0 + o, pso), si44 = next4(s_byte0 + o + ku, psi);
          u32 const ro4 = ref4(h0 + static_cast<i32>(o)), ri4 = ref4(h0 + static_cast<i32>(o + ku));
#pragma unroll
          for (u32 j = 0; j < 4; ++j) {
            qo[j] = (qo4 >> (8 * j)) & 0xFFu;
            qi[j] = (qi4 >> (8 * j)) & 0xFFu;
            so[j] = (so4 >> (8 * j)) & 0xFFu;
            si4[j] = (si44 >> (8 * j)) & 0xFFu;
            ro[j] = (ro4 >> (8 * j)) & 0xFFu;
            ri[j] = (ri4 >> (8 * j)) & 0xFFu;
          }
        }
        f64 pl[4], pi[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          pl[j] = l_phred[qo[j]];
          pi[j] = l_phred[qi[j]];
        }
        u32 wd[4] = {0, 0, 0, 0};
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
          u32 const oo = o + j;
          if (oo < o_end) {
            // floor(prefix[o+k] - prefix[o]) == 0  <=>  difference < 1.0 (prefix is non-decreasing)
            bool const errfree = (lead - lag) < 1.0;
            u32 word = (errfree ? kInstErrFree : 0u) | (oo + 1 == nk ? kInstLast : 0u);
            eacc |= (errfree ? 1u : 0u) << (oo & 31u);
            if (use_hint && mm == 0) {  // FAST: byte-identical to the reference k-mer at hint + o
              word |= static_cast<u32>(hint + static_cast<i32>(oo)) | kInstFast;
            } else {
              word |= kInstSlotMask;  // slot filled in by k_insert
              nslow++;
              sacc |= 1u << (oo & 31u);
            }
            wd[j] = word;
            if (oo + 1 < nk) {
              lag = (oo == 0) ? pl[j] : lag + pl[j];
              lead = lead + pi[j];
              if (use_hint) {
                mm -= (in_ref(oo) && so[j] == ro[j]) ? 0 : 1;
                mm += (in_ref(oo + ku) && si4[j] == ri[j]) ? 0 : 1;
              }
            }
          }
        }
        // the trip's four words in one store (a lane's words are consecutive; 64 lanes x 4 bytes per store instruction
        // was 64 separate 4-byte writes)
        if (o + 4 <= o_end) {
          u32* dst = inst_slot + ibase + o;
          if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
          } else {
            dst[0] = wd[0]; dst[1] = wd[1]; dst[2] = wd[2]; dst[3] = wd[3];
          }
        } else {
#pragma unroll
          for (u32 j = 0; j < 4; ++j)
            if (o + j < o_end) inst_slot[ibase + o + j] = wd[j];
        }
        if (((o + 4) & 31u) == 0 || o + 4 >= o_end) {  // (o is a multiple of 4: a mask word fills up exactly at a trip's end)
          mw[o >> 5] = sacc;
          ew[o >> 5] = eacc;
          sacc = eacc = 0;
        }
      }
    }
  }
  KPROF(9);
  // append this tile's slow instances to the window's queue: one atomic per wave
  u32 inc = nslow;
  for (int d = 1; d < 64; d <<= 1) {
    u32 const y = __shfl_up(inc, d);
    if (lane >= d) inc += y;
  }
  u32 const wave_total = __shfl(inc, 63);
  u32 qbase = 0;
  if (lane == 0 && wave_total) {
    qbase = atomicAdd(&ws.n_slow[a], wave_total);
    atomic_max_lazy(max_slow, qbase + wave_total);
  }
  qbase = __shfl(qbase, 0);
  u32 at = qbase + inc - nslow;
  if (nslow) {
    u32 const* mw = l_mask + static_cast<u32>(lane) * MW;
    u32 const* ew = l_emask + static_cast<u32>(lane) * MW;
    for (u32 x = my_begin >> 5; 32 * x < my_nk; ++x) {  // (this wavefront's positions [my_begin, my_nk): whole mask words)
      u32 m = all_slow ? 0xFFFFFFFFu : mw[x];
      u32 const e = ew[x];
      if (32 * x + 32 > my_nk) m &= (1u << (my_nk - 32 * x)) - 1u;
      while (m) {
        u32 const bit = __ffs(m) - 1, o = x * 32 + bit;
        m &= m - 1;
        slowq[at++] = (s_idx << 12) | o | (((e >> bit) & 1u) ? kQErrFree : 0u) | (o + 1 == read_nk ? kQLast : 0u);
      }
    }
  }
  KPROF(10);
}

// (2) k_insert: reference k-mers, then the slow queue -- one k-mer per lane, hashed from scratch (O(k)), so
//     every lane of every wave does the same work.  A general-path k-mer that IS a reference node flags its
//     read: the group-local mate-mer reasoning of k_support would be incomplete for that group.
// node identity of the k-mer s[0, k): canonical decision + fmix64 of the polynomial hash of the canonical string.
// The reverse-complement strand is hashed as the forward hash of the reversed complemented string (one multiply
// per base instead of a running power).
// k <= 32 and nothing but upper-case A/C/G/T (every k-mer of a WGS window at the usual k): the k-mer fits a 64-bit word at
// two bits per base, and both the canonical decision (kmer.cpp:17-28) and the identity come from that word.  With base i at
// bits 2i+1:2i ("little-endian" packing F) the reverse complement read as a big-endian number is simply ~F: complementing
// a base is 3 - code, and reversing the string is what turns the little-endian packing into the big-endian reading; the
// forward strand's big-endian value is F with its 2-bit groups reversed.  Big-endian numeric order of the codes A 0 < C 1 <
// G 2 < T 3 is the lexicographic order of the strings, equal means palindrome means PLUS.  id = fmix64 of the canonical
// value: node ids are internal (they never reach an output and nothing orders by them), so this need not be the oracle's
// polynomial -- it only has to be the SAME function for every k-mer of a window, which k and the alphabet decide.  A
// 25-mer costs ~90 instructions this way instead of ~400 (25 LDS reads, 25 64-bit multiply-adds, 13 byte compares).
__device__ __forceinline__ u64 packed_kmer_id(u64 f_le, int k, bool* plus_out) {
  u64 const mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
  u64 const br = ~f_le & mask;
  u32 const lo = static_cast<u32>(f_le), hi = static_cast<u32>(f_le >> 32);
  u64 r = (static_cast<u64>(__brev(lo)) << 32) | __brev(hi);  // bit i -> bit 63 - i
  r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);  // ... and the two bits of a base back in order
  u64 const bf = r >> (64 - 2 * k);
  bool const plus = bf <= br;
  u64 const id = dev_fmix64((plus ? bf : br) + static_cast<u64>(k) * kHashP);
  *plus_out = plus;
  return id ? id : 1;
}
__device__ __forceinline__ u64 kmer_id(const u8* s, int k, bool* plus_out) {
  if (k <= 32) {
    u64 f = 0;
    bool acgt = true;
#pragma unroll 8
    for (int i = 0; i < k; ++i) {
      u32 const c = s[i];
      acgt = acgt && dev_is_acgt_upper(c);
      u32 e = (c >> 1) & 3u;  // A 0, C 1, T 2, G 3
      e ^= e >> 1;            // A 0, C 1, G 2, T 3
      f |= static_cast<u64>(e) << (2 * i);
    }
    if (acgt) return packed_kmer_id(f, k, plus_out);
  }
  // the canonical decision usually falls within the first base or two, so only the canonical strand is hashed
  bool const plus = canon_plus(s, k);
  u64 h = 0;
#pragma unroll 8  // eight independent byte loads in flight instead of one round trip per base
  for (int i = 0; i < k; ++i) {  // one loop for both strands: lanes of a wavefront disagree about the strand
    u8 const b = s[plus ? i : k - 1 - i];
    h = h * kHashP + (plus ? b : dev_complement(b));
  }
  u64 const id = dev_fmix64(h);
  *plus_out = plus;
  return id ? id : 1;
}

// The same id from the window's read bases staged in LDS as 4-bit codes (A 0, C 1, G 2, T 3, N 4; k_insert stages only
// windows whose reads hold nothing else, so code -> byte is exact).  p = position of the k-mer's first base in the staged
// stream.  Same comparisons and the same polynomial over the same bytes as kmer_id: the same id.
__device__ __forceinline__ u32 seq_code(const u32* l_seq, u32 p) { return (l_seq[p >> 3] >> (4u * (p & 7u))) & 0xFu; }
__device__ __forceinline__ u32 code_byte(u32 code) { return static_cast<u32>(0x0000004E54474341ULL >> (8u * code)) & 0xFFu; }  // "ACGTN"
__device__ __forceinline__ u32 code_comp(u32 code) { return code < 4u ? 3u - code : 4u; }
__device__ __forceinline__ u32 squeeze_nibbles(u32 x) {  // eight 4-bit codes < 4 -> sixteen bits, base j at bits 2j+1:2j
  x = (x | (x >> 2)) & 0x0F0F0F0Fu;
  x = (x | (x >> 4)) & 0x00FF00FFu;
  x = (x | (x >> 8)) & 0x0000FFFFu;
  return x;
}
__device__ __forceinline__ u64 kmer_id_lds(const u32* l_seq, u32 p, int k, bool* plus_out) {
  if (k <= 32) {  // the same packed identity as kmer_id, from five words of the staged stream
    u32 const w0 = p >> 3, sh = 4u * (p & 7u);
    u32 const v0 = l_seq[w0], v1 = l_seq[w0 + 1], v2 = l_seq[w0 + 2], v3 = l_seq[w0 + 3], v4 = l_seq[w0 + 4];  // (l_seq has slack)
    u32 a0 = __builtin_amdgcn_alignbit(v1, v0, sh), a1 = __builtin_amdgcn_alignbit(v2, v1, sh),
        a2 = __builtin_amdgcn_alignbit(v3, v2, sh), a3 = __builtin_amdgcn_alignbit(v4, v3, sh);
    // keep the first k nibbles
    u32 const k0 = static_cast<u32>(k);
    auto keep = [&](u32 x, u32 first) -> u32 {
      if (k0 <= first) return 0u;
      u32 const n = k0 - first;
      return n >= 8u ? x : (x & ((1u << (4u * n)) - 1u));
    };
    a0 = keep(a0, 0);
    a1 = keep(a1, 8);
    a2 = keep(a2, 16);
    a3 = keep(a3, 24);
    if (((a0 | a1 | a2 | a3) & 0x44444444u) == 0u) {  // no N (code 4) among them
      u64 const f = static_cast<u64>(squeeze_nibbles(a0)) | (static_cast<u64>(squeeze_nibbles(a1)) << 16) |
                    (static_cast<u64>(squeeze_nibbles(a2)) << 32) | (static_cast<u64>(squeeze_nibbles(a3)) << 48);
      return packed_kmer_id(f, k, plus_out);
    }
  }
  bool plus = true;
  int const half = (k + 1) / 2;
  for (int i = 0; i < half; ++i) {  // canonical decision (kmer.cpp:17-28): inward compare of the bytes
    u32 const f = code_byte(seq_code(l_seq, p + i));
    u32 const r = code_byte(code_comp(seq_code(l_seq, p + k - 1 - i)));
    if (f != r) {
      plus = f < r;
      break;
    }
  }
  u64 h = 0;
#pragma unroll 8
  for (int i = 0; i < k; ++i) {
    u32 const cd = seq_code(l_seq, plus ? p + i : p + k - 1 - i);
    h = h * kHashP + code_byte(plus ? cd : code_comp(cd));
  }
  u64 const id = dev_fmix64(h);
  *plus_out = plus;
  return id ? id : 1;
}

// 85 % of the slow k-mers of a window are repeats (a variant's k-mers come back in every read that carries it), and
// the table insert is bound by L2 atomic throughput (two returning atomics per instance).  So the window's distinct
// k-mers are collected in an LDS map first (id -> smallest instance); the HBM table is then sized for the DISTINCT
// k-mers (a quarter of the slots the instance count would ask for: less to initialise here, less to scan in k_rank and
// k_mm_lds), each distinct k-mer goes into it once, and a last pass hands every instance its table slot.  k-mers that
// do not fit the map (deep samples) are deferred and take the direct path with its atomics.
constexpr int kInsT = 1024;
constexpr u32 kInsProbe = 1024;       // probes before an id counts as homeless (only a map without a free entry gets there)
constexpr u32 kInsMapA = 6144, kInsStageA = 1024;  // LDS map entries (48 KB of ids + 24 KB of first instances); staged sequences
// four ASCII bases -> four 2-bit codes in the low bits of each byte, and whether all four are upper-case A/C/G/T: the byte a
// code stands for is 0x41 + 2 c0 + 6 c1 + 11 c0 c1 (A 0x41, C 0x43, G 0x47, T 0x54) -- byte lanes never carry into each other
__device__ __forceinline__ u32 swar_codes4(u32 v, bool* all_acgt) {
  u32 x = (v >> 1) & 0x03030303u;          // A 0, C 1, T 2, G 3
  x ^= (x >> 1) & 0x01010101u;             // A 0, C 1, G 2, T 3
  u32 const c0 = x & 0x01010101u, c1 = (x >> 1) & 0x01010101u;
  u32 const expect = 0x41414141u + 2u * c0 + 6u * c1 + 11u * (c0 & c1);
  *all_acgt = v == expect;
  return x;
}
// Round 5: the kernel runs in two phases that share ONE 72 KB LDS area.  Phase A stages the window's read bases and hashes
// every slow k-mer: one 16-byte record per k-mer -- id, instance index + flags, sequence -- written to HBM with coalesced
// stores; phase B builds the map from the records (each thread reads back what it wrote itself: one coalesced load per k-mer,
// no gather).  With the map and the staged bases side by side the kernel held 139 KB: one workgroup, four waves per SIMD,
// per CU -- and every phase of it is a chain of round trips that wants more waves in flight.  The records carry the instance's
// error-free / last bits (k_classify) and its canonical bit, so an instance word is WRITTEN once, with its final table slot --
// the LDS map's entry, which is what the table is copied out from -- and never read: the old passes read and rewrote every
// slow instance's word twice (two 64-byte sectors per 4-byte word).
template <u32 kInsMap, u32 kStageSeqs>
__device__ __forceinline__ void insert_window(DBatch const& b, GraphWs const& ws, int const a, unsigned char* l_area) {
  constexpr u32 kInsArea = 12u * kInsMap;  // the staged bases + sequence records, THEN the map
  constexpr u32 kSeqWords = (kInsArea - 8u * kStageSeqs) / 4u - 8u;  // read bases as 4-bit codes: 131 008 / 262 080 bases (+ 8 words of slack)
  static_assert(4u * (kSeqWords + 8u) + 8u * kStageSeqs <= kInsArea, "k_insert: the staging area must fit the map's");
  __shared__ u32 l_nmap, l_ndef, l_seq_ok;
  u32* const l_seq = reinterpret_cast<u32*>(l_area);                               // phase A: [kSeqWords + 8]
  u32* const l_rpos = reinterpret_cast<u32*>(l_area + 4u * (kSeqWords + 8u));      //          [kStageSeqs] first base of a sequence in l_seq
  u32* const l_ibase = l_rpos + kStageSeqs;                                        //          [kStageSeqs] its first instance
  u64* const l_key = reinterpret_cast<u64*>(l_area);                               // phase B: [kInsMap]
  u32* const l_min = reinterpret_cast<u32*>(l_area + 8u * kInsMap);                //          [kInsMap] smallest instance of the id
  int const w = static_cast<int>(ws.active[a]);
  int const k = win_kmer(ws, w);
  int const tcl = tbl_log2(ws);
  u64* keys = ws.tbl_key + (static_cast<size_t>(a) << tcl);
  u32* first = ws.tbl_first + (static_cast<size_t>(a) << tcl);
  u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  const u32* slowq = ws.slowq + static_cast<size_t>(a) * ws.inst_stride;
  uint4* recs = ws.slow_rec + static_cast<size_t>(a) * ws.inst_stride;  // [reference k-mers | slow queue]
  u32* ref_slot_g = ws.ref_slot + static_cast<size_t>(a) * ws.ref_stride;
  u32 const base_idx = b.read_win_off[w] + w;
  u32 const r_first = b.read_win_off[w];
  SeqInfo const rsi = seq_info(b, w, 0, k);
  u32 const nq = ws.n_slow[a];
  constexpr u32 kInstIdx = (1u << 29) - 1u;  // record word z: instance index | kQLast | kQErrFree | kQPlus
  IPROF_T0();
  // ================= phase A: one record per k-mer =================
  // The slow pass hashes ~16 k k-mers per window: straight from the read bytes that was a chain of dependent HBM round trips
  // per instance.  The window's read bases are staged in LDS first (4 bit per base, aligned 8-byte loads) whenever they fit
  // and hold nothing but A/C/G/T/N; deeper or odd windows hash from HBM.
  const u8* const seq_base = [&]() {
    const u8* const first_byte = b.read_bases + b.read_off[b.read_win_off[w]];
    return first_byte - (reinterpret_cast<uintptr_t>(first_byte) & 7u);
  }();
  u64 const seq_bytes = static_cast<u64>((b.read_bases + b.read_off[b.read_win_off[w + 1]]) - seq_base);
  u32 const ns_all = seq_count(b, w);
  bool const stage = nq > 0 && seq_bytes <= static_cast<u64>(kSeqWords) * 8u && ns_all <= kStageSeqs;
  if (threadIdx.x == 0) {
    l_nmap = l_ndef = 0;
    l_seq_ok = stage ? 1u : 0u;
  }
  __syncthreads();
  if (stage) {
    const u8* const batch_lo = b.read_bases;
    const u8* const batch_hi = b.read_bases + b.read_off[b.n_reads];
    u32 const nwords = static_cast<u32>((seq_bytes + 7u) / 8u);
    bool odd = false;
    const u8* const first_read_byte = b.read_bases + b.read_off[b.read_win_off[w]];
    auto encode_word = [&](u32 wd, uint2 v) {
      bool ok_lo, ok_hi;
      u32 const xl = swar_codes4(v.x, &ok_lo), xh = swar_codes4(v.y, &ok_hi);
      if (ok_lo && ok_hi) {  // eight upper-case A/C/G/T (all but one word in a few hundred): bytes -> nibbles
        u32 yl = (xl | (xl >> 4)) & 0x00FF00FFu, yh = (xh | (xh >> 4)) & 0x00FF00FFu;
        yl = (yl | (yl >> 8)) & 0xFFFFu;
        yh = (yh | (yh >> 8)) & 0xFFFFu;
        l_seq[wd] = yl | (yh << 16);
        return;
      }
      u32 pk = 0;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        u32 const byte = ((x < 4 ? v.x : v.y) >> (8 * (x & 3))) & 0xFFu;
        u32 const e = enc_base(static_cast<u8>(byte));
        // (anything but upper-case A/C/G/T/N: a lower-case base would come back from its code in upper case)
        odd = odd || (!(e < 4u ? dev_is_acgt_upper(byte) : byte == 'N') && static_cast<u64>(wd) * 8u + x < seq_bytes &&
                      seq_base + static_cast<size_t>(wd) * 8u + x >= first_read_byte);
        pk |= e << (4 * x);
      }
      l_seq[wd] = pk;
    };
    // aligned 8-byte words; the first / last word of the BATCH may reach outside the caller's buffer (MA_MEM_DEVICE
    // passes the caller's pointer through: no alignment or padding is promised) and is read byte by byte.  Everywhere
    // else (decided once for the window, not per word: a load inside a branch waits for the one before it) four words
    // per thread are in flight.
    bool const inside = seq_base >= batch_lo && seq_base + static_cast<size_t>(nwords) * 8u <= batch_hi;
    if (inside) {
      constexpr int kWU = 4;
      for (u32 wd0 = threadIdx.x; wd0 < nwords; wd0 += kInsT * kWU) {
        uint2 v[kWU];
#pragma unroll
        for (int u = 0; u < kWU; ++u) {
          u32 const wd = min(wd0 + u * kInsT, nwords - 1);
          v[u] = *reinterpret_cast<const uint2*>(seq_base + static_cast<size_t>(wd) * 8u);
        }
#pragma unroll
        for (int u = 0; u < kWU; ++u)
          if (wd0 + u * kInsT < nwords) encode_word(wd0 + u * kInsT, v[u]);
      }
    } else {
      for (u32 wd = threadIdx.x; wd < nwords; wd += kInsT) {
        const u8* const wp = seq_base + static_cast<size_t>(wd) * 8u;
        uint2 v;
        if (wp >= batch_lo && wp + 8 <= batch_hi) {
          v = *reinterpret_cast<const uint2*>(wp);
        } else {
          u32 lo4 = 0, hi4 = 0;
          for (int x = 0; x < 8; ++x) {
            u32 const byte = (wp + x >= batch_lo && wp + x < batch_hi) ? wp[x] : 0u;
            if (x < 4) lo4 |= byte << (8 * x); else hi4 |= byte << (8 * (x - 4));
          }
          v = make_uint2(lo4, hi4);
        }
        encode_word(wd, v);
      }
    }
    for (u32 wd = nwords + threadIdx.x; wd < nwords + 8u; wd += kInsT) l_seq[wd] = 0;  // (the packed identity reads five words)
    if (odd) l_seq_ok = 0;  // a base that is not A/C/G/T/N: the codes would not give its byte back
    for (u32 sq = 1 + threadIdx.x; sq < ns_all; sq += kInsT) {  // sequence sq >= 1 is read read_win_off[w] + sq - 1
      l_rpos[sq] = static_cast<u32>((b.read_bases + b.read_off[b.read_win_off[w] + sq - 1]) - seq_base);
      l_ibase[sq] = ws.seq_inst_base[base_idx + sq];
    }
    __syncthreads();
  }
  bool const staged = l_seq_ok != 0;
  IPROF(0);  // staging
  {  // reference k-mers (graph.cpp:264-267): instance index == reference position; recs[p]
    const u8* s = b.ref_bases + rsi.off;
    for (u32 p = threadIdx.x; p < rsi.nk; p += kInsT) {
      bool plus;
      u64 const id = kmer_id(s + p, k, &plus);
      recs[p] = make_uint4(static_cast<u32>(id), static_cast<u32>(id >> 32), p | (plus ? kQPlus : 0u) | (p + 1 == rsi.nk ? kQLast : 0u), 0u);
    }
  }
  // slow queue: recs[nk + x].  Four independent chains per thread.
  constexpr int kIU = 4;
  for (u32 x0 = threadIdx.x; x0 < nq; x0 += kInsT * kIU) {
    u32 item[kIU], inst[kIU];
    u64 off[kIU], id[kIU];
    bool live[kIU], plus[kIU];
#pragma unroll
    for (int u = 0; u < kIU; ++u) {
      u32 const x = x0 + u * kInsT;
      live[u] = x < nq;
      item[u] = live[u] ? slowq[x] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kIU; ++u) {
      u32 const s_idx = (item[u] >> 12) & kQSeqMask, o = item[u] & 0xFFFu;
      if (staged) {  // (slow instances come from reads: s_idx >= 1)
        off[u] = live[u] ? l_rpos[s_idx] + o : 0u;  // position in l_seq
        inst[u] = live[u] ? l_ibase[s_idx] + o : 0u;
      } else {
        SeqInfo const si = seq_info(b, w, live[u] ? s_idx : 0u, k);
        off[u] = si.off + o;
        inst[u] = live[u] ? ws.seq_inst_base[base_idx + s_idx] + o : 0u;
      }
    }
#pragma unroll
    for (int u = 0; u < kIU; ++u) {
      id[u] = 1;
      plus[u] = true;
      if (live[u])
        id[u] = staged ? kmer_id_lds(l_seq, static_cast<u32>(off[u]), k, &plus[u]) : kmer_id(b.read_bases + off[u], k, &plus[u]);
    }
#pragma unroll
    for (int u = 0; u < kIU; ++u) {
      if (!live[u]) continue;
      recs[rsi.nk + x0 + u * kInsT] = make_uint4(static_cast<u32>(id[u]), static_cast<u32>(id[u] >> 32),
                                                 inst[u] | (item[u] & (kQErrFree | kQLast)) | (plus[u] ? kQPlus : 0u),
                                                 (item[u] >> 12) & kQSeqMask);
    }
  }
  __threadfence_block();
  __syncthreads();  // the staged bases are dead: the area becomes the map
  IPROF(1);  // records
  // ================= phase B: the map =================
  // map entry of an id (kNoNode: no room -- the map has not an entry free within kInsProbe of the id's place, and never will)
  auto map_entry = [&](u64 id) -> u32 {
    u32 e = static_cast<u32>(id >> 32) % kInsMap;
    // (a map that is all but full -- deep samples: 40 k distinct k-mers -- turns every further id away after 64 probes: each of
    //  a deep window's 200 k instances walking 1024 occupied entries was 0.4 s of this kernel per 2048 windows.  An id turned
    //  away here may still sit further along its probe sequence, put there while the map had room: the general route looks
    //  every id up with the full limit and folds every instance into the table's first-instance minimum itself.)
    u32 const lim = *reinterpret_cast<volatile u32*>(&l_nmap) + 64u >= kInsMap ? 64u : kInsProbe;
    for (u32 probe = 0; probe < lim; ++probe) {
      u64 cur = l_key[e];
      if (cur == 0) {
        unsigned long long const old = atomicCAS(reinterpret_cast<unsigned long long*>(&l_key[e]), 0ull,
                                                 static_cast<unsigned long long>(id));
        if (old == 0ull) atomicAdd(&l_nmap, 1u);
        cur = old == 0ull ? id : old;
      }
      if (cur == id) return e;
      e = e + 1 == kInsMap ? 0u : e + 1;
    }
    return kNoNode;
  };
  auto map_find = [&](u64 id) -> u32 {  // (an id that found no room within kInsProbe probes is not found within them either)
    u32 e = static_cast<u32>(id >> 32) % kInsMap;
    for (u32 probe = 0; probe < kInsProbe; ++probe) {
      u64 const cur = l_key[e];
      if (cur == id) return e;
      if (cur == 0) return kNoNode;
      e = e + 1 == kInsMap ? 0u : e + 1;
    }
    return kNoNode;
  };
  auto const word_flags = [](u32 z) -> u32 {
    return ((z & kQPlus) ? kInstPlus : 0u) | ((z & kQErrFree) ? kInstErrFree : 0u) | ((z & kQLast) ? kInstLast : 0u);
  };
  auto const rec_id = [](uint4 const& r) -> u64 { return static_cast<u64>(r.x) | (static_cast<u64>(r.y) << 32); };
  int const CW = ws.num_samples + 2;
  u32* cnt = ws.tbl_cnt + (static_cast<size_t>(a) << tcl) * CW;
  // When every id finds room in the LDS map (always, except in deep samples) the map IS the table: slot = map entry, keys and
  // first instances are copied out with coalesced stores, and every instance word is written here and now with its final slot.
  // ONE map pass over the ids of class `cls` of `ncls` (the id's low bits; one class = everything): map from scratch, the
  // reference k-mers, then the slow queue; slot = cls * kInsMap + map entry.  Returns with l_ndef = ids that found no room.
  auto map_pass = [&](u32 ncls, u32 cls) {
    __syncthreads();
    for (u32 i = threadIdx.x; i < kInsMap; i += kInsT) {
      l_key[i] = 0;
      l_min[i] = 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0) l_nmap = 0;
    __syncthreads();
    u32 const sbase = cls * kInsMap;
    for (u32 p = threadIdx.x; p < rsi.nk; p += kInsT) {
      uint4 const rc = recs[p];
      if (ncls > 1 && (rc.x & 0xFFFFu) % ncls != cls) continue;
      u32 const e = map_entry(rec_id(rc));
      if (e != kNoNode) {
        atomicMin(&l_min[e], p);
        inst_slot[p] = (sbase + e) | word_flags(rc.z);
        ref_slot_g[p] = sbase + e;
      } else {
        atomicAdd(&l_ndef, 1u);
      }
    }
    __syncthreads();  // a later instance of a reference k-mer sees a reference position as its id's minimum
    for (u32 x0 = threadIdx.x; x0 < nq; x0 += kInsT * kIU) {
      uint4 rc[kIU];
      bool live[kIU];
#pragma unroll
      for (int u = 0; u < kIU; ++u) {
        u32 const x = x0 + u * kInsT;
        live[u] = x < nq;
        rc[u] = recs[rsi.nk + (live[u] ? x : 0u)];
      }
#pragma unroll
      for (int u = 0; u < kIU; ++u) {
        if (!live[u] || (ncls > 1 && (rc[u].x & 0xFFFFu) % ncls != cls)) continue;
        u32 const e = map_entry(rec_id(rc[u]));
        if (e != kNoNode) {
          u32 const inst = rc[u].z & kInstIdx;
          u32 const before = atomicMin(&l_min[e], inst);
          inst_slot[inst] = (sbase + e) | word_flags(rc[u].z);
          // the k-mer is also a reference k-mer <=> the smallest instance of its id is a reference position
          if ((rc[u].z & kQErrFree) && before < rsi.nk) ws.rd_flag[r_first + rc[u].w - 1] = 1;
        } else {
          atomicAdd(&l_ndef, 1u);
        }
      }
    }
    __syncthreads();
  };
  auto copy_out = [&](u32 cls) {  // the map as slots [cls * kInsMap, (cls + 1) * kInsMap) of the window's table
    for (u32 i = threadIdx.x; i < kInsMap; i += kInsT) {
      u64 const id = l_key[i];
      keys[cls * kInsMap + i] = id;
      first[cls * kInsMap + i] = id ? l_min[i] : 0x7F7F7F7Fu;  // (0x7F7F7F7F > any instance)
    }
  };
  map_pass(1, 0);
  IPROF(3);  // the map
  if (l_ndef == 0 && (1u << tcl) >= kInsMap) {  // the usual case
    copy_out(0);
    for (u32 i = threadIdx.x; i < kInsMap * CW; i += kInsT) cnt[i] = 0;
    if (threadIdx.x == 0) ws.win_nslots[a] = kInsMap;
    IPROF(4);  // table out
    return;
  }
  // More distinct k-mers than the map holds (large k: an error spoils k k-mers -- the tail of the k ladder; long reads; 2.5 kb
  // windows; deep panels: 40 k distinct k-mers): the ids are split into 2 ... 16 CLASSES and the map is filled, and copied out as
  // its own range of table slots, once per class -- the records are read again (coalesced), nothing goes through an HBM hash
  // table (a deep window's 200 k slow instances through HBM compare-and-swaps took 22 ms of a workgroup).  l_ndef counts the
  // INSTANCES that found no room, a few per distinct id: the first guess is a class per ~3700 ids at four instances per id; a
  // class that still outgrows the map doubles the number of classes.
  {
    u32 const ndef0 = l_ndef;
    u32 ncls = 2;
    while (ncls < 16u && (kInsMap + ndef0 / 4u) > ncls * (kInsMap * 6u / 10u)) ncls *= 2u;
    bool tried = false;
    while (ndef0 && ncls <= 16u && (1u << tcl) >= ncls * kInsMap) {
      tried = true;
      __syncthreads();
      if (threadIdx.x == 0) l_ndef = 0;
      for (u32 cls = 0; cls < ncls; ++cls) {
        map_pass(ncls, cls);
        if (l_ndef != 0) break;  // (uniform: read after the pass's last barrier)
        copy_out(cls);
      }
      if (l_ndef == 0) {
        for (u32 i = threadIdx.x; i < ncls * kInsMap * CW; i += kInsT) cnt[i] = 0;
        if (threadIdx.x == 0) ws.win_nslots[a] = ncls * kInsMap;
        return;
      }
      ncls *= 2u;
    }
    if (tried) {  // (the classes outgrew the map after all) the general route below continues from the state of ONE pass over everything
      __syncthreads();
      if (threadIdx.x == 0) l_ndef = 0;
      map_pass(1, 0);
    }
  }
  // ---- the HBM table: as many slots as the distinct k-mers need (the stride is sized for the busiest window; nothing
  //      re-hashes later: every stage goes through the slot in the instance word) ----
  u32 nslots;
  {
    u32 tcw = 10;
    while (tcw < static_cast<u32>(tcl) && (1u << tcw) < (l_nmap + l_ndef) * 4u / 3u + 16u) ++tcw;
    nslots = 1u << tcw;
  }
  u32 const mask = nslots - 1;
  for (u32 i = threadIdx.x; i < nslots; i += kInsT) {
    keys[i] = 0ull;
    first[i] = 0x7F7F7F7Fu;
  }
  for (u32 i = threadIdx.x; i < nslots * CW; i += kInsT) cnt[i] = 0;
  if (threadIdx.x == 0) ws.win_nslots[a] = nslots;
  // ================= general route: an HBM table (deep samples; a table stride below the map's size) =================
  __syncthreads();
  // every distinct k-mer of the map into the HBM table, once; l_min[e] becomes its slot | bit 31 "also a reference k-mer"
  for (u32 e = threadIdx.x; e < kInsMap; e += kInsT) {
    u64 const id = l_key[e];
    if (id == 0) continue;
    u32 const slot = table_insert(keys, mask, id, &ws.win_flags[w]);
    u32 const fi = l_min[e];
    if (slot == kNoNode) atomicOr(&ws.win_flags[w], 4u);  // table full: the first pass plans a quarter of a deep window's instances (assemble.hip); the retry passes plan the full table
    else first[slot] = fi;  // plain store: ids of the map never take the direct path
    l_min[e] = (slot == kNoNode ? 0x7FFFFFFFu : (slot & kInstSlotMask)) | (fi < rsi.nk ? 0x80000000u : 0u);
  }
  __syncthreads();
  // ids without room in the map go straight to the table: the direct path, reference k-mers first
  auto direct_insert = [&](u64 id, u32 inst, u32* old_first) -> u32 {
    u32 const slot = table_insert(keys, mask, id, &ws.win_flags[w]);
    *old_first = 0xFFFFFFFFu;
    if (slot == kNoNode) {
      atomicOr(&ws.win_flags[w], 4u);
      return 0u;
    }
    *old_first = atomicMin(&first[slot], inst);
    return slot & kInstSlotMask;
  };
  for (u32 p = threadIdx.x; p < rsi.nk; p += kInsT) {
    uint4 const rc = recs[p];
    u32 const e = map_find(rec_id(rc));
    u32 slot, of;
    if (e != kNoNode) {
      slot = l_min[e] & 0x7FFFFFFFu;
      if (slot == 0x7FFFFFFFu) slot = 0u; else atomicMin(&first[slot], p);
    } else {
      slot = direct_insert(rec_id(rc), p, &of);
    }
    inst_slot[p] = word_flags(rc.z) | slot;
    ref_slot_g[p] = slot;
  }
  __syncthreads();  // a later instance of a reference k-mer sees a reference position as the minimum
  for (u32 x = threadIdx.x; x < nq; x += kInsT) {
    uint4 const rc = recs[rsi.nk + x];
    u32 const inst = rc.z & kInstIdx;
    u32 const e = map_find(rec_id(rc));
    u32 slot;
    bool isref;
    if (e != kNoNode) {
      u32 const sv = l_min[e];
      slot = sv & 0x7FFFFFFFu;
      if (slot == 0x7FFFFFFFu) slot = 0u; else atomicMin(&first[slot], inst);
      isref = (sv >> 31) != 0;
    } else {
      u32 of;
      slot = direct_insert(rec_id(rc), inst, &of);
      isref = of < rsi.nk;
    }
    inst_slot[inst] = slot | word_flags(rc.z);
    if ((rc.z & kQErrFree) && isref) ws.rd_flag[r_first + rc.w - 1] = 1;
  }
  IPROF(5);  // general route
}
__global__ __launch_bounds__(kInsT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_insert(DBatch b, GraphWs ws) {
  __shared__ __align__(16) unsigned char l_area[12u * kInsMapA];
  insert_window<kInsMapA, kInsStageA>(b, ws, static_cast<int>(blockIdx.x), l_area);
}
#ifdef MA_PROFILE
extern "C" void ma_debug_iprof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_iprof), sizeof(unsigned long long) * 32); }
#endif

constexpr u32 kGrSlots = 6144;   // table slots per window k_graph takes (= k_insert's direct map; a multiple of 1024)
// edge-queue entry: instance index (22 bits) and the two instance words cut down to what names their node -- FAST bit + 20 bits
// of table slot / reference position (k_graph only takes windows of at most 8192 slots and 2^18 instances)
__device__ __forceinline__ u32 edge_word(u32 word) { return ((word & kInstFast) ? (1u << 20) : 0u) | (word & 0xFFFFFu); }
__device__ __forceinline__ uint2 edge_pack(u32 ii, u32 wa, u32 wb) {
  u32 const ca = edge_word(wa), cb = edge_word(wb);
  return make_uint2((ii & 0x3FFFFFu) | (ca << 22), (ca >> 10) | (cb << 11));
}
// (3) k_support: read support of the FAST instances (node.cpp:18-24 + graph.h:102-117), one thread per group
//     of adjacent reads with equal (qname, role, sample); per-reference-position counters live in LDS.
constexpr u32 kSupCache = 2048;  // reads per window whose records k_support keeps in LDS
// eight wavefronts per window: a wavefront's turn per group of mates is a chain of LDS look-ups and a round trip for the
// instance words, ~80 groups in a row with four waves -- twice the waves, half the chain; and the workgroup's LDS is what
// the counters and the cache need (40 KB: four workgroups, 32 waves per CU; with the old mapping's mask area it was 48 KB
// and, at four waves each, 12 waves per CU)
constexpr int kSupT = 512;
__global__ __launch_bounds__(kSupT) void k_support(DBatch b, GraphWs ws, u32* max_gen, u32 cache_cap, u32 xs_log2) {
  extern __shared__ unsigned char lds_build[];
  __shared__ u32 xs_flag;
  __shared__ u32 any_big;    // some group of mates holds more k-mers than a wavefront's dedup table takes
  __shared__ u32 gen_count;  // instances routed through the general mate-mer set (sizes that set)
  __shared__ u32 genq_n;     // queue mode: keys written to the window's general-instance queue so far
  __shared__ u32 edgeq_n;    // (k+1)-mers written to the window's edge queue so far
  __shared__ u32 n_leaders;
  int const a = blockIdx.x;
  int const w = static_cast<int>(ws.active[a]);
  int const k = win_kmer(ws, w);
  int const S = ws.num_samples, CW = S + 2;
  u32* gcnt = ws.tbl_cnt + (static_cast<size_t>(a) << tbl_log2(ws)) * CW;
  u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  const u32* ref_slot_g = ws.ref_slot + static_cast<size_t>(a) * ws.ref_stride;
  u32 const ns = seq_count(b, w);
  u32 const base_idx = b.read_win_off[w] + w;
  u32 off = 0;
  u32* l_cnt = reinterpret_cast<u32*>(lds_build + off);  // packed u16 counters [ref_stride][CW] (a position is covered by < 65536 reads)
  u32 const cnt_words = (ws.ref_stride * CW + 1u) / 2u;
  off += 4u * cnt_words;
  u32* l_dd = reinterpret_cast<u32*>(lds_build + off);   // [kSupT / 64][1 << dd_log2] per-wavefront dedup tables (below)
  off += 4u * (kSupT / 64u) << ws.dd_log2;
  u32* l_mask = reinterpret_cast<u32*>(lds_build + off);
  off += 4u * (ws.max_reads + 2u);  // (the group leaders)
  // (qname, role) table of check (X): a power of two above the busiest window's read count (1024 entries up to 1022 reads)
  u32 const xs_cap = 1u << xs_log2, xs_shift = 32u - xs_log2;
  u32* l_xkey = reinterpret_cast<u32*>(lds_build + off);
  off += 4u * xs_cap;
  u32* l_xgrp = reinterpret_cast<u32*>(lds_build + off);
  off += 4u * xs_cap;
  u8* l_xsmp = lds_build + off;
  off = (off + xs_cap + 15u) & ~15u;
  // Per-sequence records for the group loop below (it was 95 % of this kernel: every group re-read flags, names, samples,
  // offsets, hints and instance bases of its reads from HBM, ten dependent round trips per group).
  //   c_meta: flags (bits 0-2) | general-path-hit flag of k_insert (bit 3) | sample << 8 | k-mer count << 16
  u32* c_meta = reinterpret_cast<u32*>(lds_build + off);
  u32* c_qn = c_meta + cache_cap;
  u32* c_ib = c_qn + cache_cap;
  i32* c_hint = reinterpret_cast<i32*>(c_ib + cache_cap);
  bool const cached = cache_cap != 0 && ns <= cache_cap;
  SeqInfo const rsi = seq_info(b, w, 0, k);
  for (u32 i = threadIdx.x; i < cnt_words; i += kSupT) l_cnt[i] = 0;
  for (u32 i = threadIdx.x; i < (kSupT / 64u) << ws.dd_log2; i += kSupT) l_dd[i] = 0;
  for (u32 i = threadIdx.x; i < xs_cap; i += kSupT) {
    l_xkey[i] = 0;
    l_xgrp[i] = 0;
  }
  if (threadIdx.x == 0) {
    xs_flag = ns > 65535u ? 1u : 0u;  // (the packed position counters hold 16 bits: such a window takes the general route)
    any_big = 0;
    gen_count = 0;
    genq_n = 0;
    edgeq_n = 0;
  }
  if (cached) {
    for (u32 sx = 1 + threadIdx.x; sx < ns; sx += kSupT) {
      u32 const r = b.read_win_off[w] + sx - 1;
      SeqInfo const si = seq_info(b, w, sx, k);
      c_meta[sx] = (b.read_flags[r] & 7u) | (ws.rd_flag[r] ? 8u : 0u) | (static_cast<u32>(b.read_sample[r]) << 8) | (min(si.nk, 0xFFFFu) << 16);
      c_qn[sx] = b.read_qname_id[r];
      c_ib[sx] = ws.seq_inst_base[base_idx + sx];
      c_hint[sx] = b.read_hint ? b.read_hint[r] : 0;
    }
  }
  __syncthreads();
  bool const hints = b.read_hint != nullptr && rsi.len <= ws.max_ref_len + 8;
  // (X) every (qname, role) key must belong to ONE sample and to ONE run of adjacent reads; otherwise the
  //     adjacent-group shortcut is not the whole story and every group goes through the general set.
  if (hints) {
    for (u32 s_idx = 1 + threadIdx.x; s_idx < ns; s_idx += kSupT) {
      u32 const r = b.read_win_off[w] + s_idx - 1;
      if (!(b.read_flags[r] & MA_RF_PASS)) continue;
      u32 const key = ((b.read_qname_id[r] << 1) | ((b.read_flags[r] & MA_RF_CASE) ? 1u : 0u)) + 1u;
      u32 h = (key * 2654435761u) >> xs_shift;
      bool done = false;
      for (u32 probe = 0; probe < xs_cap && !done; ++probe) {
        u32 cur = l_xkey[h];
        if (cur == 0) {
          u32 const old = atomicCAS(&l_xkey[h], 0u, key);
          if (old == 0) {
            l_xsmp[h] = b.read_sample[r];
            done = true;
            break;
          }
          cur = old;
        }
        if (cur == key) done = true; else h = (h + 1) & (xs_cap - 1);
      }
      if (!done) xs_flag = 1;  // table full: be conservative
    }
  }
  __syncthreads();
  if (hints) {
    for (u32 s_idx = 1 + threadIdx.x; s_idx < ns; s_idx += kSupT) {
      u32 const r = b.read_win_off[w] + s_idx - 1;
      if (!(b.read_flags[r] & MA_RF_PASS)) continue;
      bool const leader = !(s_idx > 1 && same_group(b, r, r - 1));
      u32 const key = ((b.read_qname_id[r] << 1) | ((b.read_flags[r] & MA_RF_CASE) ? 1u : 0u)) + 1u;
      u32 h = (key * 2654435761u) >> xs_shift;
      for (u32 probe = 0; probe < xs_cap; ++probe) {
        u32 const cur = l_xkey[h];
        if (cur == key) {
          if (l_xsmp[h] != b.read_sample[r]) xs_flag = 1;
          if (leader && atomicAdd(&l_xgrp[h], 1u) != 0) xs_flag = 1;  // same name in two separate runs
          break;
        }
        if (cur == 0) break;
        h = (h + 1) & (xs_cap - 1);
      }
    }
  }
  // (k_mm_lds takes a window of more than kSeqCap sequences in chunks cut between two runs of one (qname, role): no run
  //  may be longer than kMmRunMax there -- same adjacency test as its leader walk)
  if (hints && ns > kSeqCap) {
    for (u32 s_idx = 1 + kMmRunMax + threadIdx.x; s_idx < ns; s_idx += kSupT) {
      u32 const r = b.read_win_off[w] + s_idx - 1;
      u32 j = 0;
      while (j < kMmRunMax && same_group(b, r - j, r - j - 1)) ++j;
      if (j == kMmRunMax) xs_flag = 1;
    }
  }
  __syncthreads();
  bool const all_generic = !hints || xs_flag != 0;
  // Round 5, QUEUE MODE (every window whose (qname, role) keys are runs of adjacent reads and whose sequences fit an 11-bit
  // leader index -- all but deep panels and windows without hints): a general instance's mate-mer KEY (table slot << 11 |
  // run leader) is appended to the window's queue right here, where its word is in a register anyway; k_mm_q builds the LDS set
  // from the queue.  k_mm_lds found the general instances by streaming all of the window's instance words again (311 KB per
  // window for 31 KB of keys) and needed a GEN bit written back into each of them (a scattered 4-byte store per instance).
  // The queue is the slow queue's memory: k_insert is done with it.
  bool const qmode = !all_generic && !ws.mm_force_hbm && ns <= kSeqCap && tbl_log2(ws) <= 20;
  u32* const genq = ws.slowq + static_cast<size_t>(a) * ws.inst_stride;
  // DEDUP MODE.  A key names its run's leader, and a run is one group of mates -- handled by ONE wavefront, right here.  So
  // the set over the window's keys is a union of per-group sets, and a group's set (its reads' general instances, a few
  // dozen, at most the k-mers of two reads) fits a table of the wavefront's own: 512 entries tagged with the group's number
  // (no clearing between groups).  The first instance of a (table slot, group) emits slot << 4 | sample << 1 | role, and
  // what the queue then holds is one entry per COUNT: k_graph adds them up in LDS while it reads the table -- no window-wide
  // set (20 k keys per window into a 16 k / 32 k-entry LDS set: the slowest kernel of the build stage), no kernel of its own.
  // For the windows k_graph takes (ws.graph_fused, a table of at most kGrSlots slots) whose groups all fit their table.
  u32 const dd_cap = 1u << ws.dd_log2;
  // EDGE QUEUE: the (k+1)-mers of the reads that are not the reference's own edges (both k-mers FAST at consecutive reference
  // positions: 7 of 8) -- instance index and both words packed into 8 bytes (edge_pack), into the memory k_insert's records lived in.  k_graph
  // builds the window's edges from this queue and the reference's own words: after k_classify has written the instance words
  // this kernel is the only one that streams them.
  uint2* const edgeq = reinterpret_cast<uint2*>(ws.slow_rec + static_cast<size_t>(a) * ws.inst_stride);

  // One WAVEFRONT per group of mates, a lane per k-mer: the instance words of a read are read coalesced, the offsets
  // of the first mate that were counted are one ballot per 64 k-mers (kept in scalar registers), and the second mate
  // looks its reference position up in them.  (A thread per group walked its reads' words one by one: serial,
  // uncoalesced, 250 dependent loads per thread.)
  // leaders of the groups, in any order (counting commutes)
  u32* const l_leaders = l_mask;  // [max_reads + 2] (the per-thread masks of the old mapping lived here)
  if (threadIdx.x == 0) n_leaders = 0;
  __syncthreads();
  for (u32 s_idx = 1 + threadIdx.x; s_idx < ns; s_idx += kSupT) {
    u32 const r0 = b.read_win_off[w] + s_idx - 1;
    if (!(b.read_flags[r0] & MA_RF_PASS)) continue;
    if (s_idx > 1 && same_group(b, r0, r0 - 1)) continue;  // not the leader
    l_leaders[atomicAdd(&n_leaders, 1u)] = s_idx;
    if (qmode) {  // k-mers of the whole group against the dedup table's capacity (3/4 full at most)
      u32 tot = 0;
      for (u32 j = 0; s_idx + j < ns && (j == 0 || same_group(b, r0, r0 + j)); ++j) tot += seq_info(b, w, s_idx + j, k).nk;
      if (tot > dd_cap / 4u * 3u) any_big = 1;
    }
  }
  __syncthreads();
  bool const qdedup = qmode && ws.graph_fused && !any_big && ws.win_nslots[a] <= kGrSlots;
  IPROF_T0();
  u32 const wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  u32* const dd = l_dd + (wave << ws.dd_log2);
  u32 wave_gen = 0;
  // (sequence sa, sb of this window in one group: same_group() on the cached records)
  auto same_group_c = [&](u32 sa, u32 sb) {
    u32 const ma_ = c_meta[sa], mb = c_meta[sb];
    return (mb & MA_RF_PASS) && c_qn[sa] == c_qn[sb] && ((ma_ ^ mb) & MA_RF_CASE) == 0 && ((ma_ ^ mb) & 0xFF00u) == 0;
  };
  for (u32 gi = wave; gi < n_leaders; gi += kSupT / 64) {
    u32 const s_idx = l_leaders[gi];
    u32 const r0 = b.read_win_off[w] + s_idx - 1;
    u32 gsize = 1;
    if (cached) {
      while (s_idx + gsize < ns && same_group_c(s_idx, s_idx + gsize)) gsize++;
    } else {
      while (s_idx + gsize < ns && same_group(b, r0, r0 + gsize)) gsize++;
    }
    bool generic = all_generic || gsize > 2;
    for (u32 gm = 0; gm < gsize && !generic; ++gm) {
      if (cached) {
        u32 const mt = c_meta[s_idx + gm];
        if ((mt & 8u) || (mt >> 16) > 32u * kMaskWords) generic = true;
      } else {
        SeqInfo const si = seq_info(b, w, s_idx + gm, k);
        if (ws.rd_flag[r0 + gm] || si.nk > 32u * kMaskWords) generic = true;
      }
    }
    // the instance words of both mates in flight at once (reads of up to 128 k-mers with cached records): one round trip
    // per group instead of one per 64 k-mers of each mate
    u32 pre00 = 0, pre01 = 0, pre10 = 0, pre11 = 0;
    bool have_pre = cached && gsize <= 2;
    for (u32 gm = 0; gm < gsize && have_pre; ++gm) {
      u32 const nkc = c_meta[s_idx + gm] >> 16;
      if (nkc == 0xFFFFu || nkc > 128u) have_pre = false;
    }
    if (have_pre) {
      u32 const nka = c_meta[s_idx] >> 16, iba = c_ib[s_idx];
      pre00 = lane < nka ? inst_slot[iba + lane] : 0u;
      pre01 = 64u + lane < nka ? inst_slot[iba + 64u + lane] : 0u;
      if (gsize == 2) {
        u32 const nkb = c_meta[s_idx + 1] >> 16, ibb = c_ib[s_idx + 1];
        pre10 = lane < nkb ? inst_slot[ibb + lane] : 0u;
        pre11 = 64u + lane < nkb ? inst_slot[ibb + 64u + lane] : 0u;
      }
    }
    i32 hint0 = 0;
    u32 nk0 = 0;
    unsigned long long m0[kMaskWords / 2];  // offsets of the first mate that were counted
#pragma unroll
    for (int x = 0; x < kMaskWords / 2; ++x) m0[x] = 0;
    for (u32 gm = 0; gm < gsize; ++gm) {
      u32 const sx = s_idx + gm;
      u32 const r = r0 + gm;
      SeqInfo si;
      u32 ibase, smp, role;
      i32 hint_r;
      if (cached && (c_meta[sx] >> 16) != 0xFFFFu) {
        u32 const mt = c_meta[sx];
        si.nk = mt >> 16;
        ibase = c_ib[sx];
        smp = (mt >> 8) & 0xFFu;
        role = (mt & MA_RF_CASE) ? 1u : 0u;
        hint_r = c_hint[sx];
      } else {
        si = seq_info(b, w, sx, k);
        ibase = ws.seq_inst_base[base_idx + sx];
        smp = b.read_sample[r];
        role = (b.read_flags[r] & MA_RF_CASE) ? 1u : 0u;
        hint_r = hints ? b.read_hint[r] : 0;
      }
      if (si.nk == 0) continue;
      if (smp >= static_cast<u32>(S)) smp = S - 1;
      if (gm == 0) {
        hint0 = hints ? hint_r : 0;
        nk0 = si.nk;
      }
      for (u32 ob = 0; ob < si.nk; ob += 64) {
        u32 const o = ob + lane;
        u32 const word = have_pre ? (gm == 0 ? (ob == 0 ? pre00 : pre01) : (ob == 0 ? pre10 : pre11))
                                  : (o < si.nk ? inst_slot[ibase + o] : 0u);
        {
          u32 const dn = __shfl_down(word, 1, 64);
          u32 first_next;  // the word at offset ob + 64 (the neighbour of lane 63's)
          if (have_pre) first_next = ob == 0 ? __shfl(gm == 0 ? pre01 : pre11, 0, 64) : 0u;
          else first_next = ob + 64 < si.nk ? inst_slot[ibase + ob + 64] : 0u;
          u32 const nextw = lane == 63 ? first_next : dn;
          bool const pair = o + 1 < si.nk && !((word & kInstFast) && (nextw & kInstFast) &&
                                               (nextw & kInstSlotMask) == (word & kInstSlotMask) + 1u);
          unsigned long long const em = __ballot(pair);
          if (em) {
            u32 eb = 0;
            if (lane == 0) eb = atomicAdd(&edgeq_n, static_cast<u32>(__popcll(em)));
            eb = __shfl(eb, 0, 64);
            if (pair) edgeq[eb + static_cast<u32>(__popcll(em & ((1ull << lane) - 1ull)))] = edge_pack(ibase + o, word, nextw);
          }
        }
        bool const ef = (word & kInstErrFree) != 0;
        bool const to_gen = ef && (generic || !(word & kInstFast));
        u32 const p = word & kInstSlotMask;
        unsigned long long const gmask = __ballot(to_gen);
        if (qdedup) {
          if (gmask) {
            u32 const nslot = to_gen ? ((word & kInstFast) ? ref_slot_g[p] : p) : 0u;
            bool won = false;
            if (to_gen) {  // first of its (table slot, group)?  entry = slot << 11 | tag; an entry with another tag is free
              u32 const tag = gi + 1u, mine = (nslot << 11) | tag;
              u32 h = (nslot * 2654435761u) >> (32u - ws.dd_log2);
              for (;;) {
                u32 cur = dd[h];
                if ((cur & 0x7FFu) != tag) {
                  u32 const old = atomicCAS(&dd[h], cur, mine);
                  if (old == cur) {
                    won = true;
                    break;
                  }
                  cur = old;
                  if ((cur & 0x7FFu) != tag) continue;  // (not reachable: only this group's lanes write the table now)
                }
                if (cur == mine) break;  // counted already
                h = (h + 1u) & (dd_cap - 1u);
              }
            }
            unsigned long long const wm = __ballot(won);
            if (wm) {
              u32 qb = 0;
              if (lane == 0) qb = atomicAdd(&genq_n, static_cast<u32>(__popcll(wm)));
              qb = __shfl(qb, 0, 64);
              if (won) genq[qb + static_cast<u32>(__popcll(wm & ((1ull << lane) - 1ull)))] = (nslot << 4) | (smp << 1) | role;
            }
          }
        } else if (qmode) {
          if (gmask) {
            u32 qb = 0;
            if (lane == 0) qb = atomicAdd(&genq_n, static_cast<u32>(__popcll(gmask)));
            qb = __shfl(qb, 0, 64);
            if (to_gen) {
              u32 const nslot = (word & kInstFast) ? ref_slot_g[p] : p;
              genq[qb + static_cast<u32>(__popcll(gmask & ((1ull << lane) - 1ull)))] = ((nslot << 11) | (s_idx - 1u)) + 1u;
            }
          }
        } else if (to_gen) {
          inst_slot[ibase + o] = word | kInstGen;  // exact handling by the mate-mer set kernels
        }
        wave_gen += static_cast<u32>(__popcll(gmask));
        bool const fast = ef && !to_gen;
        bool dup = false;
        if (gm == 1 && nk0 > 0 && fast) {  // did the first member count this reference position?
          i64 const o0 = static_cast<i64>(p) - hint0;
          if (o0 >= 0 && o0 < static_cast<i64>(nk0)) {
            unsigned long long wd = 0;
#pragma unroll
            for (int x = 0; x < kMaskWords / 2; ++x) wd = (o0 >> 6) == x ? m0[x] : wd;
            dup = (wd >> (o0 & 63)) & 1ull;
          }
        }
        if (gm == 0 && !generic) {
          unsigned long long const cm = __ballot(fast);
#pragma unroll
          for (int x = 0; x < kMaskWords / 2; ++x)
            if (static_cast<int>(ob >> 6) == x) m0[x] = cm;
        }
        if (fast && !dup) {
          u32 const i1 = p * CW + smp, i2 = p * CW + S + role;
          atomicAdd(&l_cnt[i1 >> 1], 1u << ((i1 & 1u) * 16u));
          atomicAdd(&l_cnt[i2 >> 1], 1u << ((i2 & 1u) * 16u));
        }
      }
    }
  }
  if (lane == 0 && wave_gen) atomicAdd(&gen_count, wave_gen);
  IPROF_SYNC(27);  // group loop
  if (threadIdx.x == 0) {
    // Windows whose (qname, role) keys each map to ONE run of adjacent reads (xs_flag == 0) and whose general
    // instances fit the LDS set are finished by k_mm_lds; the others need the HBM-resident set (max_gen[1]).
    // (up to four passes over the set for a window that fits the LDS tables; a deeper one goes chunk by chunk)
    bool const lds_ok = !all_generic && !ws.mm_force_hbm && (ns > kSeqCap || gen_count <= 4u * kMmLdsMax) && tbl_log2(ws) <= 20;
    ws.n_edgeq[a] = edgeq_n;
    ws.n_genq[a] = (qmode || qdedup) ? genq_n : 0u;
    // bit 29: counts queued for k_graph (dedup mode), bit 30: keys queued for k_mm_q, bit 31: HBM set; else k_mm_lds
    ws.mm_mode[a] = qdedup ? (gen_count | 0x20000000u) : qmode ? (gen_count | 0x40000000u) : (gen_count | (lds_ok ? 0u : 0x80000000u));
    atomic_max_lazy(max_gen, gen_count);
    ws.mm_log2[a] = 0;
    if (!qmode && !lds_ok && gen_count) {
      // the window's HBM-resident set: 12 bytes per entry at a load of 3/4, carved out of the chunk's pool.  A pool that is
      // used up is a capacity like any other: the window is flagged and the retry pass, whose pool holds a full set per
      // window, re-assembles it.
      atomic_max_lazy(max_gen + 1, gen_count);
      u32 lg = 10;
      while (lg < 31u && (1ull << lg) < static_cast<u64>(gen_count) * 4u / 3u + 16u) ++lg;
      unsigned long long const bytes = ((12ull << lg) + 255ull) & ~255ull;
      unsigned long long const at = atomicAdd(ws.mm_pool_used, bytes);
      if (at + bytes <= ws.mm_pool_bytes) {
        ws.mm_off[a] = at;
        ws.mm_log2[a] = lg;
      } else {
        atomicOr(&ws.win_flags[w], 4u);
      }
    }
  }
  for (u32 i = threadIdx.x; i < rsi.nk * CW; i += kSupT) {
    u32 const v = (l_cnt[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
    if (v == 0) continue;
    u32 const p = i / CW, x = i % CW;
    atomicAdd(&gcnt[static_cast<size_t>(ref_slot_g[p]) * CW + x], v);
  }
  IPROF(28);  // flush of the position counters
}

__device__ __forceinline__ u64 mm_key_of(u32 slot, u32 qname, u32 role) {
  return ((static_cast<u64>(slot) << 33) | (static_cast<u64>(qname) << 1) | role) + 1ull;
}
__device__ __forceinline__ u32 inst_table_slot(u32 word, const u32* ref_slot_g) {
  return (word & kInstFast) ? ref_slot_g[word & kInstSlotMask] : (word & kInstSlotMask);
}

// k_mm_insert / k_count / k_rank / k_edges walk the window's instance words in FLAT order (lane i reads
// word base + i: coalesced); the read of an instance is recovered from the per-sequence instance bases
// only for the few instances that need it.
__device__ __forceinline__ void mm_hbm_insert(DBatch const& b, GraphWs const& ws, int a, u32* l_base) {
  if (!(ws.mm_mode[a] & 0x80000000u) || ws.mm_log2[a] == 0) return;  // done by k_mm_lds / no set (flagged)
  int const w = static_cast<int>(ws.active[a]);
  u32 const mlg = ws.mm_log2[a];
  u32 const mask = (1u << mlg) - 1;
  u64* keys = reinterpret_cast<u64*>(ws.mm_pool + ws.mm_off[a]);
  u32* mins = reinterpret_cast<u32*>(keys + (size_t(1) << mlg));
  const u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  const u32* ref_slot_g = ws.ref_slot + static_cast<size_t>(a) * ws.ref_stride;
  u32 const ns = seq_count(b, w);
  u32 const base_idx = b.read_win_off[w] + w;
  const u32* sbase = stage_seq_bases(ws.seq_inst_base + base_idx, ns, l_base);
  // the set is the workgroup's own: it clears it itself (16-byte stores; the region is 256-byte aligned)
  for (u32 i = threadIdx.x; i < (1u << mlg) / 2u; i += kBT) reinterpret_cast<uint4*>(keys)[i] = make_uint4(0, 0, 0, 0);
  for (u32 i = threadIdx.x; i < (1u << mlg) / 4u; i += kBT)
    reinterpret_cast<uint4*>(mins)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  __threadfence_block();
  __syncthreads();
  u32 const ninst = ws.win_ninst[w], nref = ninst - ws.win_nread_inst[w];
  for (u32 ii = nref + threadIdx.x; ii < ninst; ii += kBT) {
    u32 const v = inst_slot[ii];
    if ((v & (kInstErrFree | kInstGen)) != (kInstErrFree | kInstGen)) continue;
    u32 const s_idx = seq_of(sbase, ns, ii);
    u32 const r = b.read_win_off[w] + s_idx - 1;
    u32 const qn = b.read_qname_id[r];
    u32 const role = (b.read_flags[r] & MA_RF_CASE) ? 1u : 0u;
    u64 const key = mm_key_of(inst_table_slot(v, ref_slot_g), qn, role);
    u32 slot = static_cast<u32>(dev_fmix64(key)) & mask;
    for (u32 probe = 0; probe <= mask; ++probe) {
      u64 cur = keys[slot];
      if (cur != key && cur == 0) {
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[slot]), 0ull,
                                           static_cast<unsigned long long>(key));
        cur = (old == 0ull) ? key : old;
      }
      if (cur == key) {
        atomicMin(&mins[slot], ii);
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

__device__ __forceinline__ void mm_hbm_count(DBatch const& b, GraphWs const& ws, int a, u32* l_base) {
  if (!(ws.mm_mode[a] & 0x80000000u) || ws.mm_log2[a] == 0) return;  // done by k_mm_lds / no set (flagged)
  int const w = static_cast<int>(ws.active[a]);
  int const S = ws.num_samples, CW = S + 2;
  u32 const mlg = ws.mm_log2[a];
  u32 const mask = (1u << mlg) - 1;
  const u64* keys = reinterpret_cast<const u64*>(ws.mm_pool + ws.mm_off[a]);
  const u32* mins = reinterpret_cast<const u32*>(keys + (size_t(1) << mlg));
  u32* cnt = ws.tbl_cnt + (static_cast<size_t>(a) << tbl_log2(ws)) * CW;
  const u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  const u32* ref_slot_g = ws.ref_slot + static_cast<size_t>(a) * ws.ref_stride;
  u32 const ns = seq_count(b, w);
  u32 const base_idx = b.read_win_off[w] + w;
  const u32* sbase = stage_seq_bases(ws.seq_inst_base + base_idx, ns, l_base);
  __syncthreads();
  u32 const ninst = ws.win_ninst[w], nref = ninst - ws.win_nread_inst[w];
  for (u32 ii = nref + threadIdx.x; ii < ninst; ii += kBT) {
    u32 const v = inst_slot[ii];
    if ((v & (kInstErrFree | kInstGen)) != (kInstErrFree | kInstGen)) continue;
    u32 const s_idx = seq_of(sbase, ns, ii);
    u32 const r = b.read_win_off[w] + s_idx - 1;
    u32 const qn = b.read_qname_id[r];
    u32 const role = (b.read_flags[r] & MA_RF_CASE) ? 1u : 0u;
    u32 sample = b.read_sample[r];
    if (sample >= static_cast<u32>(S)) sample = S - 1;
    u32 const nslot = inst_table_slot(v, ref_slot_g);
    u64 const key = mm_key_of(nslot, qn, role);
    u32 slot = static_cast<u32>(dev_fmix64(key)) & mask;
    for (u32 probe = 0; probe <= mask; ++probe) {
      u64 const cur = keys[slot];
      if (cur == key) break;
      if (cur == 0) {
        slot = kNoNode;
        break;
      }
      slot = (slot + 1) & mask;
    }
    if (slot == kNoNode || mins[slot] != ii) continue;  // a previous (qname, role, kmer) wins
    atomicAdd(&cnt[static_cast<size_t>(nslot) * CW + sample], 1u);
    atomicAdd(&cnt[static_cast<size_t>(nslot) * CW + S + role], 1u);
  }
}

// the HBM-resident mate-mer set of the windows that need one (no mapping hints, names in separate runs, capacity retries):
// build it, then count through it -- one workgroup per such window does both
__global__ __launch_bounds__(kBT) void k_mm_hbm(DBatch b, GraphWs ws) {
  __shared__ u32 l_base[kSeqCap];
  int const a = blockIdx.x;
  if (!(ws.mm_mode[a] & 0x80000000u) || ws.mm_log2[a] == 0) return;  // done by k_mm_q / k_mm_lds, or no set (flagged)
  mm_hbm_insert(b, ws, a, l_base);
  __threadfence_block();
  __syncthreads();
  mm_hbm_count(b, ws, a, l_base);
}

// k_mm_lds: the general mate-mer set (graph.h:102-117) and its read support (node.cpp:18-24) for one window, entirely
// in LDS.  k_mm_insert / k_count keep a (slot, qname, role) set per window in HBM: ~12 k random 128-byte-line
// accesses per window and pass (profiles/: 8-10 GB of traffic per launch for ~0.6 GB of useful bytes).  When every
// (qname, role) key of the window is one run of adjacent reads (k_support checked that: xs_flag == 0), the run's
// leader index is a dense 11-bit name, so key = slot << 11 | leader fits 32 bits and a 32 k-entry set fits in LDS.
// The key also determines what is counted -- sample and role are those of the leader -- so "the first instance of
// a key counts" needs no minimum: one coalesced walk builds the set, the support is then read off the set.
constexpr int kMmT = 1024;
constexpr u32 kMmQueue = 4096;
constexpr u32 kMmAux = kSeqCap + kSeqCap / 2 + kMmQueue;  // 28 KB
template <bool kTestProbe>  // (tests: the set counts as full after ws.mm_probe_max probes)
__global__ __launch_bounds__(kMmT) void k_mm_lds(DBatch b, GraphWs ws) {
  __shared__ u32 l_set[kMmLdsCap];
  __shared__ u32 l_aux[kMmAux];  // walk: instance bases | run leaders | queue; afterwards: packed u16 support counters
  __shared__ u32 l_qn;
  // sequence of every 128th instance: an instance's sequence is then a table look-up and a step or two along the instance
  // bases instead of an 11-step binary search (a chain of dependent LDS reads per general instance, ~20 k per window)
  constexpr u32 kBlkShift = 7, kBlkCap = 1024;  // windows of up to 128 Ki instances
  __shared__ u16 l_blk[kBlkCap];
  u32* const l_base = l_aux;
  u16* const l_lead = reinterpret_cast<u16*>(l_aux + kSeqCap);
  u32* const l_queue = l_aux + kSeqCap + kSeqCap / 2;
  int const a = blockIdx.x;
  u32 const mode = ws.mm_mode[a];
  if ((mode & 0xE0000000u) || mode == 0) return;  // HBM set / queued for k_mm_q or k_graph / nothing to do
  int const w = static_cast<int>(ws.active[a]);
  int const S = ws.num_samples, CW = S + 2;
  u32* cnt = ws.tbl_cnt + (static_cast<size_t>(a) << tbl_log2(ws)) * CW;
  const u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  const u32* ref_slot_g = ws.ref_slot + static_cast<size_t>(a) * ws.ref_stride;
  u32 const ns = seq_count(b, w);
  u32 const base_idx = b.read_win_off[w] + w;
  u32 const r_base = b.read_win_off[w];
  // Windows with more general instances than one set holds (deep samples) take several passes, pass p handling the
  // table slots with slot % npass == p: every key lives in exactly one pass, so each pass is complete in itself.
  // Windows with more sequences than the LDS tables hold (deep panels: 7 k reads) go through them in CHUNKS of up to
  // kSeqCap - 1 consecutive sequences cut between two runs: a key names its run's leader, so the keys of two chunks are
  // disjoint and every chunk is complete in itself as well (k_support: no run longer than kMmRunMax in such a window).
  __shared__ u32 l_chunk[4];  // s_lo, s_hi of the chunk, its general instances
  constexpr u32 kMmWork = 48;
  __shared__ u32 l_wl[kMmWork], l_wn, l_witem, l_sfull;
  u32 const ninst = ws.win_ninst[w], nref = ninst - ws.win_nread_inst[w];
  bool const chunked = ns > kSeqCap;
  IPROF_T0();
  for (u32 s_lo = 0; s_lo < ns;) {
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 e = ns;
    if (chunked) {
      e = min(s_lo + kSeqCap - 1u - kMmRunMax, ns);
      while (e < ns && same_group(b, r_base + e - 1, r_base + e - 2)) ++e;  // (e >= 2 here)
    }
    l_chunk[0] = e;
    l_chunk[1] = 0;
  }
  __syncthreads();
  u32 const s_hi = l_chunk[0], nsc = s_hi - s_lo, s_one = max(s_lo, 1u);
  u32 const i_lo = s_lo == 0 ? 0u : ws.seq_inst_base[base_idx + s_lo];
  u32 const i_hi = s_hi < ns ? ws.seq_inst_base[base_idx + s_hi] : ninst;
  u32 cgen = mode;
  if (chunked) {  // the chunk's general instances (sizes its passes)
    u32 mine = 0;
    for (u32 ii = max(i_lo, nref) + threadIdx.x; ii < i_hi; ii += kMmT)
      mine += (inst_slot[ii] & (kInstErrFree | kInstGen)) == (kInstErrFree | kInstGen);
    for (u32 o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&l_chunk[1], mine);
    __syncthreads();
    cgen = l_chunk[1];
  }
  // The chunk's slot classes (modulus << 16 | residue), a work list: a class whose keys fill the set -- the passes are sized
  // by an instance COUNT, but the keys of a k-mer that a thousand read pairs carry all fall into its slot's class -- is split
  // in two and redone before anything of it was counted.  (Round 4 flagged such a window and the retry pass re-assembled it
  // from scratch with an HBM-resident set: two of the bench's 512 deep-panel windows, a fifth of that leg's step.)
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 const np0 = min(max((cgen + kMmLdsMax - 1u) / kMmLdsMax, 1u), 16u);
    for (u32 p = 0; p < np0; ++p) l_wl[p] = (np0 << 16) | p;
    l_wn = cgen ? np0 : 0u;
  }
  __syncthreads();
  u32 chunk_items = 0;
  for (;;) {
  __syncthreads();
  if (threadIdx.x == 0) {
    l_sfull = 0;
    l_witem = l_wn ? l_wl[--l_wn] : 0u;
  }
  __syncthreads();
  u32 const witem = l_witem;
  if (witem == 0) break;
  bool const first_item = chunk_items++ == 0;  // (the chunk's block table is built once)
  u32 const npass = witem >> 16, pass = witem & 0xFFFFu;
  for (u32 i = threadIdx.x; i < kMmLdsCap; i += kMmT) l_set[i] = 0;
  for (u32 sr = threadIdx.x; sr < nsc; sr += kMmT) {
    u32 const s = s_lo + sr;
    l_base[sr] = ws.seq_inst_base[base_idx + s];
    u32 lead = s;  // sequence s >= 1 is read r_base + s - 1; its run's leader
    while (lead > s_one && same_group(b, r_base + lead - 1, r_base + lead - 2)) --lead;
    l_lead[sr] = static_cast<u16>(lead - min(lead, s_one));  // relative to the chunk's first read
  }
  __syncthreads();
  u32 blk_shift = kBlkShift;
  while (blk_shift < 10 && i_hi - i_lo > (kBlkCap << blk_shift)) ++blk_shift;  // (a chunk of 2 k reads: 258 k instances)
  bool const blk_ok = i_hi - i_lo <= (kBlkCap << blk_shift);
  if (blk_ok && first_item)
    for (u32 bk = threadIdx.x; (bk << blk_shift) < i_hi - i_lo; bk += kMmT)
      l_blk[bk] = static_cast<u16>(seq_of(l_base, nsc, i_lo + (bk << blk_shift)));
  __syncthreads();
  IPROF(12);  // set init, instance bases, run leaders
  // (qname, role, node) of one general instance -> set
  bool set_full = false;
  auto const visit_slot = [&](u32 ii, u32 nslot) {
    u32 sq;
    if (blk_ok) {
      sq = l_blk[(ii - i_lo) >> blk_shift];
      while (sq + 1 < nsc && l_base[sq + 1] <= ii) ++sq;
    } else {
      sq = seq_of(l_base, nsc, ii);
    }
    u32 const lead = l_lead[sq];
    if (npass > 1 && nslot % npass != pass) return;
    u32 const key = ((nslot << 11) | lead) + 1u;
    u32 h = (key * 2654435761u) >> 17;  // kMmLdsCap == 1 << 15
    for (u32 probe = 0; probe < (kTestProbe ? ws.mm_probe_max : kMmLdsCap); ++probe) {
      u32 cur = l_set[h];
      if (cur == 0) {
        u32 const old = atomicCAS(&l_set[h], 0u, key);
        cur = old == 0 ? key : old;
      }
      if (cur == key) {
        if constexpr (kTestProbe) return;
        break;
      }
      h = (h + 1) & (kMmLdsCap - 1);
    }
    // No room.  The passes are sized by an instance COUNT, but the keys of a k-mer that a thousand read pairs carry all fall
    // into its slot's pass: a pass can outgrow the set.  Never silent: the window is flagged like any other capacity, the
    // retry pass re-assembles it from scratch and sends its mate-mers through the HBM set (k_support: mm_force_hbm).
    // (the product kernel probes the whole set: a key finds no room only when NO entry is free -- seen below, for nothing,
    //  as nkeys == kMmLdsCap; the test variant gives up after mm_probe_max probes and says so here)
    if constexpr (kTestProbe) set_full = true;
  };
  auto const visit = [&](u32 ii, u32 word) { visit_slot(ii, inst_table_slot(word, ref_slot_g)); };
  // Only ~20 % of the instances are general ones: visiting them where they are found keeps 4 of 5 lanes idle
  // through a chain of dependent LDS / L2 accesses.  Each chunk of 16 k words is therefore scanned with
  // coalesced loads first, the general instances are queued (one LDS atomic per wavefront), and the queue is
  // then visited with full wavefronts.
  constexpr int kU = 4;           // independent loads in flight per thread
  constexpr u32 kChunk = kMmT * 16;
  u32 const lane = threadIdx.x & 63u;
  for (u32 c0 = max(i_lo, nref); c0 < i_hi; c0 += kChunk) {
    if (threadIdx.x == 0) l_qn = 0;
    __syncthreads();
    u32 const c1 = min(c0 + kChunk, i_hi);
    for (u32 i0 = c0 + threadIdx.x; i0 < c1 + kMmT * kU; i0 += kMmT * kU) {  // whole wavefronts reach the ballots
      u32 v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        u32 const ii = i0 + u * kMmT;
        v[u] = ii < c1 ? inst_slot[ii] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        bool const gen = (v[u] & (kInstErrFree | kInstGen)) == (kInstErrFree | kInstGen);
        u64 const m = __ballot(gen);
        if (m == 0) continue;
        u32 qb = 0;
        if (lane == 0) qb = atomicAdd(&l_qn, static_cast<u32>(__popcll(m)));
        qb = __shfl(qb, 0, 64);
        if (gen) {
          u32 const at = qb + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)));
          if (at < kMmQueue) l_queue[at] = i0 + u * kMmT;
          else visit(i0 + u * kMmT, v[u]);  // queue full (more than a quarter of the chunk): visit in place
        }
      }
    }
    __syncthreads();
    u32 const qn = min(l_qn, kMmQueue);
    // (four queue entries per thread in flight: instance word, then -- for an instance on a reference k-mer -- its table
    //  slot, were two dependent round trips per entry)
    for (u32 q0 = threadIdx.x; q0 < qn; q0 += kMmT * kU) {
      u32 qi[kU], qw[kU], qs[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        u32 const q = q0 + u * kMmT;
        qi[u] = q < qn ? l_queue[q] : 0xFFFFFFFFu;
        qw[u] = qi[u] != 0xFFFFFFFFu ? inst_slot[qi[u]] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) qs[u] = qi[u] != 0xFFFFFFFFu ? inst_table_slot(qw[u], ref_slot_g) : 0u;
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (qi[u] != 0xFFFFFFFFu) visit_slot(qi[u], qs[u]);
    }
    __syncthreads();
  }
  // Read support: one count per key, for the leader's sample and role.  Random global atomics would cost a
  // 128-byte HBM line each (measured: 6 of this kernel's 8 ms), so the counts are accumulated in LDS as packed
  // u16 (a slot has at most 2047 names) for a range of slots at a time and added to the window's table in index
  // order: a wavefront's 64 counters share two lines.
  // The set is not probed any more: its keys are compacted to the front (every thread reads its 32 entries into
  // registers first, so compacting in place is safe) and the rest of the set joins l_aux as counter space -- three
  // or four slot ranges instead of ten, each scanning only the keys.
  IPROF(13);  // scan + queue + set inserts
  if (kTestProbe && set_full) l_sfull = 1;
  u32 nkeys = 0;
  {
    constexpr u32 kPer = kMmLdsCap / kMmT;  // 32 consecutive entries per thread
    u32 mine[kPer];
    u32 cntk = 0;
#pragma unroll
    for (u32 x = 0; x < kPer; ++x) {
      mine[x] = l_set[threadIdx.x * kPer + x];
      cntk += mine[x] != 0;
    }
    __syncthreads();
    // exclusive scan of cntk over the 1024 threads: wave scan + wave totals in l_aux
    u32 inc = cntk;
#pragma unroll
    for (u32 o = 1; o < 64; o <<= 1) {
      u32 const y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
    }
    if (lane == 63) l_aux[threadIdx.x >> 6] = inc;
    __syncthreads();
    u32 woff = 0, total = 0;
    for (u32 k = 0; k < kMmT / 64; ++k) {
      u32 const t = l_aux[k];
      if (k < (threadIdx.x >> 6)) woff += t;
      total += t;
    }
    __syncthreads();
    u32 at = woff + inc - cntk;
#pragma unroll
    for (u32 x = 0; x < kPer; ++x)
      if (mine[x] != 0) l_set[at++] = mine[x];
    nkeys = total;
    __syncthreads();
  }
  if (nkeys == kMmLdsCap || l_sfull) {  // not an entry free (or, in the test build, a probe sequence cut short): keys may have found no room
    if (npass < 8192u) {
      if (threadIdx.x == 0 && l_wn + 2u <= kMmWork) {
        l_wl[l_wn++] = ((2u * npass) << 16) | pass;
        l_wl[l_wn++] = ((2u * npass) << 16) | (pass + npass);
      } else if (threadIdx.x == 0) {
        atomicOr(&ws.win_flags[w], 4u);
      }
    } else if (threadIdx.x == 0) {
      atomicOr(&ws.win_flags[w], 4u);  // one table slot alone holds more keys than the set: the retry pass's HBM set takes the window
    }
    continue;
  }
  IPROF(14);  // key compaction
  // counter space: l_aux, then the free tail of l_set (u32 words, two u16 counters each)
  u32 const tail_words = kMmLdsCap - nkeys;
  u32 const ctr_words = kMmAux + tail_words;
  auto const ctr = [&](u32 wd) -> u32& { return wd < kMmAux ? l_aux[wd] : l_set[nkeys + (wd - kMmAux)]; };
  u32 const slots_per_pass = (2u * ctr_words) / static_cast<u32>(CW);
  u32 const tcap = ws.win_nslots[a];
  for (u32 s0 = 0; s0 < tcap; s0 += slots_per_pass) {
    for (u32 i = threadIdx.x; i < ctr_words; i += kMmT) ctr(i) = 0;
    __syncthreads();
    for (u32 i = threadIdx.x; i < nkeys; i += kMmT) {
      u32 const key = l_set[i];
      u32 const nslot = (key - 1u) >> 11;
      if (nslot < s0 || nslot - s0 >= slots_per_pass) continue;
      u32 const r = r_base + s_one - 1u + ((key - 1u) & 2047u);
      u32 sample = b.read_sample[r];
      if (sample >= static_cast<u32>(S)) sample = S - 1;
      u32 const role = (b.read_flags[r] & MA_RF_CASE) ? 1u : 0u;
      u32 const i1 = (nslot - s0) * CW + sample, i2 = (nslot - s0) * CW + S + role;
      atomicAdd(&ctr(i1 >> 1), 1u << ((i1 & 1u) * 16u));
      atomicAdd(&ctr(i2 >> 1), 1u << ((i2 & 1u) * 16u));
    }
    __syncthreads();
    u32 const nc = min(slots_per_pass, tcap - s0) * CW;
    for (u32 j = threadIdx.x; j < nc; j += kMmT) {
      u32 const c = (ctr(j >> 1) >> ((j & 1u) * 16u)) & 0xFFFFu;
      if (c) atomicAdd(&cnt[static_cast<size_t>(s0) * CW + j], c);  // no return value: the wave does not wait for it
    }
    __syncthreads();
  }
  IPROF(15);  // support counters
  }  // pass
  s_lo = s_hi;
  }  // chunk
}

// k_mm_q: the mate-mer set and its read support from the window's KEY QUEUE (k_support, queue mode).  Same set, same
// compaction, same packed counters as k_mm_lds -- without the walk over the instance words, the instance bases, run leaders
// and block table that walk needed (30 KB of LDS), and in two sizes: a 16 k-entry set (76 KB: two workgroups per CU) for the
// windows with up to 14 k general instances -- the usual WGS window has 8 k --, a 32 k-entry set for the rest, which take
// as many passes over slot classes as their keys need (a pass re-reads the queue: 4 bytes per key).
template <int kCapLog2>
__device__ __forceinline__ void mm_q_window(DBatch const& b, GraphWs const& ws, int const a) {
  constexpr u32 kCap = 1u << kCapLog2, kMax = kCap * 5u / 8u;  // (a pass fills the set to 5/8: linear probing at 7/8 walks dozens of entries per key)
  constexpr u32 kAux = kCapLog2 == 14 ? 3072u : 7168u;
  __shared__ u32 l_set[kCap];
  __shared__ u32 l_aux[kAux];  // packed u16 support counters (+ the free tail of the set)
  __shared__ u32 l_full;
  u32 const mode = ws.mm_mode[a];
  if ((mode & 0xE0000000u) != 0x40000000u) return;
  u32 const gen = mode & 0x1FFFFFFFu;
  if (gen == 0) return;
  int const w = static_cast<int>(ws.active[a]);
  int const S = ws.num_samples, CW = S + 2;
  u32* cnt = ws.tbl_cnt + (static_cast<size_t>(a) << tbl_log2(ws)) * CW;
  const u32* genq = ws.slowq + static_cast<size_t>(a) * ws.inst_stride;
  u32 const r_base = b.read_win_off[w];
  u32 const lane = threadIdx.x & 63u;
  u32 const pmax = min(ws.mm_probe_max, kCap);
  u32 const npass = (gen + kMax - 1u) / kMax;
  if (threadIdx.x == 0) l_full = 0;
  IPROF_T0();
  for (u32 pass = 0; pass < npass; ++pass) {
    __syncthreads();
    for (u32 i = threadIdx.x; i < kCap; i += kMmT) l_set[i] = 0;
    IPROF_SYNC(22);  // set init
    constexpr int kU = 4;
    for (u32 q0 = threadIdx.x; q0 < gen; q0 += kMmT * kU) {
      u32 key[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) key[u] = q0 + u * kMmT < gen ? genq[q0 + u * kMmT] : 0u;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (key[u] == 0 || (npass > 1 && ((key[u] - 1u) >> 11) % npass != pass)) continue;
        u32 h = (key[u] * 2654435761u) >> (32 - kCapLog2);
        bool placed = false;
        for (u32 probe = 0; probe < pmax; ++probe) {
          u32 cur = l_set[h];
          if (cur == 0) {
            u32 const old = atomicCAS(&l_set[h], 0u, key[u]);
            cur = old == 0 ? key[u] : old;
          }
          if (cur == key[u]) {
            placed = true;
            break;
          }
          h = (h + 1) & (kCap - 1);
          if ((probe & 127u) == 127u && l_full) break;  // somebody has found the set full: the window is redone anyway
        }
        // No room.  The passes are sized by a key COUNT, but the keys of a k-mer that a thousand read pairs carry all fall into
        // its slot's pass: a pass can outgrow the set.  Never silent: the window is flagged like any other capacity, the retry
        // pass re-assembles it from scratch and sends its mate-mers through the HBM set (k_support: mm_force_hbm).
        if (!placed) l_full = 1;
      }
    }
    IPROF_SYNC(23);  // queue -> set
    // The set is not probed any more: its keys are compacted to the front (every thread reads its entries into registers
    // first, so compacting in place is safe) and the rest of the set joins l_aux as counter space.
    u32 nkeys = 0;
    {
      constexpr u32 kPer = kCap / kMmT;
      u32 mine[kPer];
      u32 cntk = 0;
#pragma unroll
      for (u32 x = 0; x < kPer; ++x) {
        mine[x] = l_set[threadIdx.x * kPer + x];
        cntk += mine[x] != 0;
      }
      __syncthreads();
      u32 inc = cntk;
#pragma unroll
      for (u32 o = 1; o < 64; o <<= 1) {
        u32 const y = __shfl_up(inc, o, 64);
        if (lane >= o) inc += y;
      }
      if (lane == 63) l_aux[threadIdx.x >> 6] = inc;
      __syncthreads();
      u32 woff = 0, total = 0;
      for (u32 k = 0; k < kMmT / 64; ++k) {
        u32 const t = l_aux[k];
        if (k < (threadIdx.x >> 6)) woff += t;
        total += t;
      }
      __syncthreads();
      u32 at = woff + inc - cntk;
#pragma unroll
      for (u32 x = 0; x < kPer; ++x)
        if (mine[x] != 0) l_set[at++] = mine[x];
      nkeys = total;
    }
    IPROF_SYNC(24);  // compaction
    // Read support: one count per key, for the leader's sample and role -- accumulated in LDS as packed u16 (a slot has at most
    // 2047 names) for a range of slots at a time and added to the window's table in index order.
    u32 const tail_words = kCap - nkeys;
    u32 const ctr_words = kAux + tail_words;
    auto const ctr = [&](u32 wd) -> u32& { return wd < kAux ? l_aux[wd] : l_set[nkeys + (wd - kAux)]; };
    u32 const slots_per_pass = (2u * ctr_words) / static_cast<u32>(CW);
    u32 const tcap = ws.win_nslots[a];
    for (u32 s0 = 0; s0 < tcap; s0 += slots_per_pass) {
      for (u32 i = threadIdx.x; i < ctr_words; i += kMmT) ctr(i) = 0;
      __syncthreads();
      for (u32 i = threadIdx.x; i < nkeys; i += kMmT) {
        u32 const key = l_set[i];
        u32 const nslot = (key - 1u) >> 11;
        if (nslot < s0 || nslot - s0 >= slots_per_pass) continue;
        u32 const r = r_base + ((key - 1u) & 2047u);  // the run's leader: sequence (key & 2047) + 1 is read r_base + that
        u32 sample = b.read_sample[r];
        if (sample >= static_cast<u32>(S)) sample = S - 1;
        u32 const role = (b.read_flags[r] & MA_RF_CASE) ? 1u : 0u;
        u32 const i1 = (nslot - s0) * CW + sample, i2 = (nslot - s0) * CW + S + role;
        atomicAdd(&ctr(i1 >> 1), 1u << ((i1 & 1u) * 16u));
        atomicAdd(&ctr(i2 >> 1), 1u << ((i2 & 1u) * 16u));
      }
      __syncthreads();
      u32 const nc = min(slots_per_pass, tcap - s0) * CW;
      for (u32 j = threadIdx.x; j < nc; j += kMmT) {
        u32 const c = (ctr(j >> 1) >> ((j & 1u) * 16u)) & 0xFFFFu;
        if (c) atomicAdd(&cnt[static_cast<size_t>(s0) * CW + j], c);  // no return value: the wave does not wait for it
      }
      __syncthreads();
    }
    IPROF(25);  // support counters
  }
  __syncthreads();
  if (l_full && threadIdx.x == 0) atomicOr(&ws.win_flags[w], 4u);
}
// (a few hundred workgroups walk the batch: since the dedup mode this is the route of the few windows whose groups of mates do
//  not fit a wavefront's table, or that k_graph does not take)
template <int kCapLog2>
__global__ __launch_bounds__(kMmT) void k_mm_q(DBatch b, GraphWs ws) {
  for (int a = blockIdx.x; a < ws.n_active; a += gridDim.x) {
    mm_q_window<kCapLog2>(b, ws, a);
    __syncthreads();
  }
}

// low-coverage pruning (graph.cpp:363-390 with component 0 == everything, no anchors yet) and
// canonical ranking of the survivors by first-insertion order.
// sixteen wavefronts per window: the table pass and the node records are chains of round trips to HBM per thread, the
// ranking scan a chain of tiles -- four times the threads, a quarter of the chain (0.48 -> 0.39 ms per 2048 windows)
constexpr int kRankT = 1024;
__device__ __forceinline__ void rank_window(DBatch const& b, GraphWs const& ws, u32 min_node_cov, int a) {
  __shared__ u32 sh[kRankT / 64];
  __shared__ u32 l_base[kSeqCap];
  int const w = static_cast<int>(ws.active[a]);
  int const S = ws.num_samples, CW = S + 2;
  u32 const tcap = ws.win_nslots[a];  // slots this window uses (k_insert)
  int const tcl = tbl_log2(ws);
  if (ws.gr_done[a]) return;  // k_graph took this window
  const u64* keys = ws.tbl_key + (static_cast<size_t>(a) << tcl);
  u32* first = ws.tbl_first + (static_cast<size_t>(a) << tcl);
  u32* slot_node = ws.slot_node + (static_cast<size_t>(a) << tcl);
  const u32* cnt = ws.tbl_cnt + (static_cast<size_t>(a) << tcl) * CW;
  u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  size_t const nb = static_cast<size_t>(a) * ws.nc;

  // 1. survivor flag into bit 31 of tbl_first (set = pruned / empty); a survivor marks its first instance word, so
  //    that step 2 streams the instance words instead of gathering tbl_first once per instance
  // (four slots per thread in flight: key -> counts -> first instance -> its word is a chain of four round trips)
  constexpr int kRU = 4;
  for (u32 s0 = threadIdx.x; s0 < tcap; s0 += kRankT * kRU) {
    u64 ky[kRU];
    u32 fi[kRU];
    bool live[kRU], remove[kRU];
#pragma unroll
    for (int u = 0; u < kRU; ++u) {
      u32 const s = s0 + u * kRankT;
      live[u] = s < tcap;
      u32 const sc = live[u] ? s : 0u;  // (every load of the trip unconditional: guarded loads go out one at a time)
      ky[u] = keys[sc];
      fi[u] = first[sc];
      u32 total = 0;
      bool any = false, all = true;
      for (int i = 0; i < S; ++i) {
        u32 const c = cnt[static_cast<size_t>(sc) * CW + i];
        total += c;
        any |= c > 0;
        all &= c <= 1;
      }
      remove[u] = (any && all) || total < min_node_cov;  // node.cpp:38-42, graph.cpp:374-378
    }
#pragma unroll
    for (int u = 0; u < kRU; ++u) {
      u32 const s = s0 + u * kRankT;
      if (!live[u]) continue;
      slot_node[s] = kNoNode;
      if (ky[u] == 0)
        first[s] = 0xFFFFFFFFu;
      else if (remove[u])
        first[s] = fi[u] | 0x80000000u;
      else
        atomicOr(&inst_slot[fi[u]], kInstFirst);  // (one slot per instance: no other thread touches this word; no value to wait for)
    }
  }
  __syncthreads();

  // 2. canonical ranking: node index = number of "first instance of a surviving node" events before it in
  //    instance order.  Four consecutive instance words per thread and tile (one 16-byte load), block scan.
  u32 const ns = seq_count(b, w);
  u32 const base_idx = b.read_win_off[w] + w;
  const u32* sbase = stage_seq_bases(ws.seq_inst_base + base_idx, ns, l_base);
  __syncthreads();
  u32 const ninst = ws.win_ninst[w];
  u32 const read_byte0 = static_cast<u32>(0);
  (void)read_byte0;
  u64 const win_read_off0 = b.read_off[b.read_win_off[w]];
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 running = 0;
  // 2a. the ranks: 16 consecutive instance words per thread and tile (four 16-byte loads, the next tile's in flight while
  //     this one is scanned), block scan; a first instance only leaves its index in the window's node list here.  (Writing
  //     the node's record on the spot -- sequence search, two dependent loads of read fields, the counters -- kept the
  //     whole workgroup at the tile's barrier for three round trips to HBM, 76 tiles in a row.)
  u32* node_inst = ws.scratch + nb * 32;  // [nc] instance index of node idx (clean-stage scratch: free during the build)
  constexpr u32 kWT = 16;
  u32 const last4 = ninst ? ((ninst - 1) & ~3u) : 0u;
  auto load16 = [&](u32 ii0, uint4* v) {
#pragma unroll
    for (u32 q = 0; q < kWT / 4; ++q) v[q] = *reinterpret_cast<const uint4*>(inst_slot + min(ii0 + 4 * q, last4));  // (clamped: masked below)
  };
  uint4 nxt[kWT / 4];
  load16(kWT * threadIdx.x, nxt);
  for (u32 tile0 = 0; tile0 < ninst; tile0 += kWT * kRankT) {
    u32 const ii0 = tile0 + kWT * threadIdx.x;
    uint4 cur[kWT / 4];
#pragma unroll
    for (u32 q = 0; q < kWT / 4; ++q) cur[q] = nxt[q];
    if (tile0 + kWT * kRankT < ninst) load16(ii0 + kWT * kRankT, nxt);
    u32 fmask = 0;
#pragma unroll
    for (u32 q = 0; q < kWT / 4; ++q) {
      u32 const vv[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
      for (u32 j2 = 0; j2 < 4; ++j2) {
        u32 const ii = ii0 + 4 * q + j2, v = vv[j2];
        if (ii < ninst && !(v & kInstFast) && (v & kInstFirst)) fmask |= 1u << (4 * q + j2);
      }
    }
    u32 const mine = __popc(fmask);
    u32 inc = mine;
    for (int d = 1; d < 64; d <<= 1) {
      u32 const y = __shfl_up(inc, d);
      if (lane >= d) inc += y;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    u32 before = 0, tile_total = 0;
    for (int x = 0; x < kRankT / 64; ++x) {
      u32 const t = sh[x];
      if (x < wave) before += t;
      tile_total += t;
    }
    __syncthreads();
    u32 idx = running + before + inc - mine;
    running += tile_total;
    for (u32 m = fmask; m; m &= m - 1) {
      if (idx < ws.nc) node_inst[idx] = ii0 + static_cast<u32>(__ffs(m)) - 1u;
      idx++;
    }
  }
  __threadfence_block();
  __syncthreads();
  // 2b. the node records, a thread per node: independent chains, as many in flight as there are threads
  {
    u32 const nn = min(running, ws.nc);
    for (u32 idx = threadIdx.x; idx < nn; idx += kRankT) {
      u32 const ii = node_inst[idx];
      u32 const v = inst_slot[ii];
      u32 const slot = v & kInstSlotMask;
      u32 const s = seq_of(sbase, ns, ii);
      u32 const o = ii - sbase[s];
      u32 label = 1;  // Label::REFERENCE
      u32 srcbit = 0, rel_off = 0;
      if (s > 0) {
        u32 const r = b.read_win_off[w] + s - 1;
        label = (b.read_flags[r] & MA_RF_CASE) ? 4u : 2u;  // Label::CASE / Label::CTRL
        srcbit = 0x80000000u;
        rel_off = static_cast<u32>(b.read_off[r] - win_read_off0);
      }
      slot_node[slot] = idx;
      for (int i = 0; i < S; ++i) ws.nd_cnt[(nb + idx) * S + i] = cnt[static_cast<size_t>(slot) * CW + i];
      ws.nd_role[(nb + idx) * 2 + 0] = cnt[static_cast<size_t>(slot) * CW + S];
      ws.nd_role[(nb + idx) * 2 + 1] = cnt[static_cast<size_t>(slot) * CW + S + 1];
      ws.nd_src[nb + idx] = srcbit | (rel_off + o);
      ws.nd_label[nb + idx] = static_cast<u8>(label);
      ws.nd_sign[nb + idx] = (v & kInstPlus) ? 1 : 0;
      ws.nd_nedge[nb + idx] = 0;
    }
  }
  u32 const total = running;
  if (threadIdx.x == 0) ws.n_nodes[a] = total;
  if (total >= ws.nc) {  // capacity exceeded: flagged, window fails
    if (threadIdx.x == 0) atomicOr(&ws.win_flags[w], 4u);
    return;
  }
  for (u32 x = threadIdx.x; x < total * kEdgeCap; x += kRankT) {
    ws.nd_edge[nb * kEdgeCap + x] = 0xFFFFFFFFu;
    ws.nd_ekey[nb * kEdgeCap + x] = 0xFFFFFFFFu;
  }
  __syncthreads();
  // 3. mRefNodeIds (graph.cpp:264-267): node of every reference k-mer (kNoNode when pruned)
  SeqInfo const rsi = seq_info(b, w, 0, win_kmer(ws, w));
  u32* refn = ws.ref_node + static_cast<size_t>(a) * ws.ref_stride;
  for (u32 p = threadIdx.x; p < ws.ref_stride; p += kRankT)
    refn[p] = p < rsi.nk ? slot_node[inst_slot[p] & kInstSlotMask] : kNoNode;
}

__device__ __forceinline__ void edge_insert(GraphWs const& ws, size_t nb, u32 node, u32 val, u32 key, u32* flags) {
  u32* ed = ws.nd_edge + (nb + node) * kEdgeCap;
  u32* ek = ws.nd_ekey + (nb + node) * kEdgeCap;
  for (int e = 0; e < kEdgeCap; ++e) {
    u32 cur = ed[e];
    if (cur == 0xFFFFFFFFu) {
      u32 const old = atomicCAS(&ed[e], 0xFFFFFFFFu, val);
      cur = (old == 0xFFFFFFFFu) ? val : old;
    }
    if (cur == val) {
      atomicMin(&ek[e], key);
      return;
    }
  }
  atomicOr(flags, 4u);  // more than kEdgeCap distinct edges at one node
}

// k_edges: an edge and its mirror per (k+1)-mer (graph.cpp:333-337).  ~7 of 8 read (k+1)-mers are the reference's
// own edges and are skipped; the rest repeat a few thousand distinct edges many times.  Inserting every occurrence
// into the per-node 16-slot lists in HBM cost a chain of dependent random line accesses each (measured: 8 ms,
// ~60 GB of traffic per launch).  The window's distinct directed edges are therefore collected in an LDS set
// first -- key = src << 17 | dst << 2 | kind, value = smallest order key (first occurrence) -- and each of them
// is then stored once, its slot taken from a per-node LDS counter.  k_edge_sort orders the slots by order key.
constexpr int kEdT = 1024;
constexpr u32 kEdgeSet = 8192;    // 64 KB of keys + order keys and 8 KB of counters: TWO workgroups per CU (16 k entries made it
constexpr u32 kEdgeNodes = 8192;  // one, and the lanes' copies of this kernel queued behind each other); fuller windows insert directly
constexpr u32 kEdgeEmpty = 0xFFFFFFFFu;
__device__ __forceinline__ void edges_window(DBatch const& b, GraphWs const& ws, int a) {
  __shared__ u32 l_key[kEdgeSet];
  __shared__ u32 l_min[kEdgeSet];
  __shared__ u8 l_deg[kEdgeNodes];
  __shared__ u32 l_fail;
  int const w = static_cast<int>(ws.active[a]);
  if ((ws.win_flags[w] & 4u) || ws.gr_done[a]) return;
  const u32* slot_node = ws.slot_node + (static_cast<size_t>(a) << tbl_log2(ws));
  const u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  const u32* refn = ws.ref_node + static_cast<size_t>(a) * ws.ref_stride;
  size_t const nb = static_cast<size_t>(a) * ws.nc;
  u32 const n_nodes = min(ws.n_nodes[a], ws.nc);
  bool const use_set = ws.nc <= kEdgeNodes;  // per-node slot counters in LDS; 15-bit node indices in the packed key
  for (u32 i = threadIdx.x; i < kEdgeSet; i += kEdT) {
    l_key[i] = kEdgeEmpty;
    l_min[i] = 0xFFFFFFFFu;
  }
  for (u32 i = threadIdx.x; i < kEdgeNodes / 4u; i += kEdT) reinterpret_cast<u32*>(l_deg)[i] = 0;
  if (threadIdx.x == 0) l_fail = use_set ? 0u : 1u;
  __syncthreads();
  // one (k+1)-mer per lane in flat instance order: (ii, ii + 1) unless ii is the last k-mer of its sequence
  u32 const ninst = ws.win_ninst[w], nref = ninst - ws.win_nread_inst[w];
  u32 const last = ninst > 0 ? ninst - 1 : 0;
  auto const set_insert = [&](u32 src, u32 val, u32 okey) {
    u32 const key = (src << 17) | val;
    if (key == kEdgeEmpty) {
      l_fail = 1;
      return;
    }
    u32 h = (key * 2654435761u) >> 19;  // kEdgeSet == 1 << 13
    for (u32 probe = 0; probe < 256u; ++probe) {
      u32 cur = l_key[h];
      if (cur == kEdgeEmpty) {
        u32 const old = atomicCAS(&l_key[h], kEdgeEmpty, key);
        cur = old == kEdgeEmpty ? key : old;
      }
      if (cur == key) {
        atomicMin(&l_min[h], okey);
        return;
      }
      h = (h + 1) & (kEdgeSet - 1);
    }
    l_fail = 1;  // set (nearly) full: the window is redone with direct insertion below
  };
  // returns false for (k+1)-mers that add nothing
  auto const edge_of = [&](u32 ii, u32 wa, u32 wb, u32* na, u32* nbn, u32* fwd, u32* rev) -> bool {
    if (wa & kInstLast) return false;
    // both k-mers are reference nodes at consecutive positions: this (k+1)-mer is the reference's own edge,
    // already inserted (with a smaller order key) by the reference sequence itself
    if (ii >= nref && (wa & kInstFast) && (wb & kInstFast) && (wb & kInstSlotMask) == (wa & kInstSlotMask) + 1) return false;
    *na = (wa & kInstFast) ? refn[wa & kInstSlotMask] : slot_node[wa & kInstSlotMask];
    *nbn = (wb & kInstFast) ? refn[wb & kInstSlotMask] : slot_node[wb & kInstSlotMask];
    if (*na == kNoNode || *nbn == kNoNode) return false;
    // edge kind from the STORED signs of both nodes (graph.cpp:333-336)
    u32 const sa_minus = ws.nd_sign[nb + *na] ? 0u : 1u, sb_minus = ws.nd_sign[nb + *nbn] ? 0u : 1u;
    *fwd = (sa_minus << 1) | sb_minus;                 // MakeFwdEdgeKind(sA, sB)
    *rev = ((sb_minus ^ 1u) << 1) | (sa_minus ^ 1u);   // RevEdgeKind(fwd) seen from B
    return true;
  };
  constexpr int kU = 4;
  if (use_set) {
    for (u32 i0 = threadIdx.x; i0 < last; i0 += kEdT * kU) {
      u32 wa[kU], wb[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        u32 const ii = i0 + u * kEdT;
        wa[u] = ii < last ? inst_slot[ii] : kInstLast;
        wb[u] = ii < last ? inst_slot[ii + 1] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        u32 const ii = i0 + u * kEdT;
        u32 na, nbn, fwd, rev;
        if (!edge_of(ii, wa[u], wb[u], &na, &nbn, &fwd, &rev)) continue;
        set_insert(na, (nbn << 2) | fwd, 2u * ii);
        set_insert(nbn, (na << 2) | rev, 2u * ii + 1u);
      }
    }
  }
  __syncthreads();
  if (l_fail == 0) {
    for (u32 i = threadIdx.x; i < kEdgeSet; i += kEdT) {
      u32 const key = l_key[i];
      if (key == kEdgeEmpty) continue;
      u32 const src = key >> 17, val = key & 0x1FFFFu;
      // one u8 counter per node, four to a word
      u32 const old = atomicAdd(reinterpret_cast<u32*>(l_deg) + (src >> 2), 1u << ((src & 3u) * 8u));
      u32 const slot = (old >> ((src & 3u) * 8u)) & 0xFFu;
      if (slot >= static_cast<u32>(kEdgeCap) || src >= n_nodes) {
        atomicOr(&ws.win_flags[w], 4u);  // more than kEdgeCap distinct edges at one node
        continue;
      }
      ws.nd_edge[(nb + src) * kEdgeCap + slot] = val;
      ws.nd_ekey[(nb + src) * kEdgeCap + slot] = l_min[i];
    }
    return;
  }
  // direct insertion (set unusable for this window)
  for (u32 ii = threadIdx.x; ii < last; ii += kEdT) {
    u32 na, nbn, fwd, rev;
    if (!edge_of(ii, inst_slot[ii], inst_slot[ii + 1], &na, &nbn, &fwd, &rev)) continue;
    edge_insert(ws, nb, na, (nbn << 2) | fwd, 2u * ii, &ws.win_flags[w]);
    edge_insert(ws, nb, nbn, (na << 2) | rev, 2u * ii + 1u, &ws.win_flags[w]);
  }
}

__device__ __forceinline__ void edge_sort_window(GraphWs const& ws, int a) {
  size_t const nb = static_cast<size_t>(a) * ws.nc;
  u32 const n = min(ws.n_nodes[a], ws.nc);
  for (u32 i = threadIdx.x; i < n; i += kEdT) {
    u32* ed = ws.nd_edge + (nb + i) * kEdgeCap;
    u32* ek = ws.nd_ekey + (nb + i) * kEdgeCap;
    u32 v[kEdgeCap], kk[kEdgeCap];
    int m = 0;
    // (four slots of both arrays per load: a node has two edges as a rule -- one round trip instead of three)
    uint4 e4 = *reinterpret_cast<const uint4*>(ed), k4 = *reinterpret_cast<const uint4*>(ek);
    for (int e = 0; e < kEdgeCap; ++e) {
      if ((e & 3) == 0 && e > 0) {
        e4 = *reinterpret_cast<const uint4*>(ed + e);
        k4 = *reinterpret_cast<const uint4*>(ek + e);
      }
      u32 const cv = (e & 3) == 0 ? e4.x : ((e & 3) == 1 ? e4.y : ((e & 3) == 2 ? e4.z : e4.w));
      u32 const ck = (e & 3) == 0 ? k4.x : ((e & 3) == 1 ? k4.y : ((e & 3) == 2 ? k4.z : k4.w));
      if (cv == 0xFFFFFFFFu) break;
      // insertion sort by first-occurrence key (EmplaceEdge order, node.h:59-64)
      int j = m++;
      while (j > 0 && kk[j - 1] > ck) {
        kk[j] = kk[j - 1];
        v[j] = v[j - 1];
        --j;
      }
      kk[j] = ck;
      v[j] = cv;
    }
    for (int e = 0; e < m; ++e) ed[e] = v[e];
    ws.nd_nedge[nb + i] = static_cast<u8>(m);
  }
}

// the general route of the graph records: low-coverage pruning + ranking, edges, edge order from the table and the instance
// words in HBM -- for the windows k_graph does not take (a table beyond 8192 slots: deep panels, 2.5 kb windows; an edge set
// that filled up).  A few hundred workgroups walk the batch's windows and skip the ones marked done.
static_assert(kRankT == kEdT, "k_graph_gen runs the three phases with one workgroup size");
__global__ __launch_bounds__(kRankT) void k_graph_gen(DBatch b, GraphWs ws, u32 min_node_cov) {
  for (int a = blockIdx.x; a < ws.n_active; a += gridDim.x) {
    if (ws.gr_done[a]) continue;  // k_graph took this window
    rank_window(b, ws, min_node_cov, a);
    __threadfence_block();
    __syncthreads();
    edges_window(b, ws, a);
    __threadfence_block();
    __syncthreads();
    edge_sort_window(ws, a);
    __syncthreads();
  }
}

// k_graph: k_rank + k_edges + k_edge_sort for the common window in ONE workgroup-per-window kernel that streams the
// instance words ONCE (the three kernels streamed them twice, marked first instances in them with atomics, and went through
// HBM for the slot -> node map, the edge slots' 0xFF initialisation and the order keys).
//   * ranking without the instance stream: a survivor's rank is the number of survivors with a smaller FIRST instance, and
//     the table already holds every slot's first instance -- a bitmap over the instance indices (one bit per survivor's first
//     instance, 4 KB per 32 Ki instances), a prefix count per bitmap word, rank = prefix + popcount below the bit;
//   * slot -> node and reference position -> node stay in LDS (u16) for the edge pass;
//   * edges: one coalesced walk over consecutive instance words (16-byte loads); the distinct directed edges go into an LDS set
//     probed from hash(SOURCE NODE) only, so that all edges of a node sit in one run of occupied entries: an edge's place in
//     its node's list (EmplaceEdge order = order of first occurrence, node.h:59-64) is the number of that run's entries with
//     the same source and a smaller order key -- no per-node counters, no order keys in HBM, no sort kernel.
// LDS: 16 KB + 1 KB + 2 B per reference k-mer + max(ranking area, 8 B x kGrSet) = ~69 KB: two workgroups per CU.
// A window this kernel does not take (table of more than 8192 slots, an edge set that fills up) is left untouched for
// k_rank / k_edges / k_edge_sort, which skip the windows marked done here.
constexpr int kGrT = 1024;
constexpr u32 kGrSet = 6144;     // entries of the edge set
constexpr u32 kGrProbe = 384;    // longest run of occupied entries a lookup walks before the window is handed back
constexpr u32 kGrNone = 0xFFFFu;
constexpr u32 kGrEmpty = 0xFFFFFFFFu;
__host__ __device__ inline u32 graph_area_bytes(u32 inst_stride, int S) {
  u32 const nw = (inst_stride + 31u) / 32u + 1u;
  u32 const rank_area = 4u * nw + 2u * nw + 4u * kSeqCap + 64u;
  u32 const set_area = 8u * kGrSet;
  u32 const ctr_area = 2u * kGrSlots * static_cast<u32>(S + 2);
  u32 m = rank_area > set_area ? rank_area : set_area;
  m = m > ctr_area ? m : ctr_area;
  return (m + 15u) & ~15u;
}
__host__ __device__ inline u32 graph_lds_bytes(u32 inst_stride, u32 ref_stride, int S) {
  return 2u * kGrSlots + kGrSlots / 8u + ((2u * ref_stride + 15u) & ~15u) + graph_area_bytes(inst_stride, S) + 64u;
}
// (eight waves per SIMD = two workgroups per CU: 64 VGPRs)
__global__ __launch_bounds__(kGrT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_graph(DBatch b, GraphWs ws, u32 min_node_cov) {
  extern __shared__ unsigned char lds_build[];
  __shared__ u32 sh[kGrT / 64];
  __shared__ u32 l_fail;
  int const a = blockIdx.x;
  int const w = static_cast<int>(ws.active[a]);
  int const S = ws.num_samples, CW = S + 2;
  int const tcl = tbl_log2(ws);
  u32 const tcap = ws.win_nslots[a];
  u32 const ninst = ws.win_ninst[w], nref = ninst - ws.win_nread_inst[w];
  u32 const nw = (ws.inst_stride + 31u) / 32u + 1u;
  if (threadIdx.x == 0) ws.gr_done[a] = 0;
  if ((ws.win_flags[w] & 4u) || tcap > kGrSlots || ninst > ws.inst_stride || ninst >= (1u << 22)) return;  // (uniform)
  const u64* keys = ws.tbl_key + (static_cast<size_t>(a) << tcl);
  const u32* first = ws.tbl_first + (static_cast<size_t>(a) << tcl);
  u32* cnt = ws.tbl_cnt + (static_cast<size_t>(a) << tcl) * CW;
  const u32* inst_slot = ws.inst_slot + static_cast<size_t>(a) * ws.inst_stride;
  const u32* ref_slot_g = ws.ref_slot + static_cast<size_t>(a) * ws.ref_stride;
  size_t const nb = static_cast<size_t>(a) * ws.nc;
  // LDS carve
  u16* l_node = reinterpret_cast<u16*>(lds_build);                        // [kGrSlots] slot -> node
  u32* l_sign = reinterpret_cast<u32*>(lds_build + 2u * kGrSlots);        // [kGrSlots / 32] node -> stored sign is PLUS
  u16* l_refn = reinterpret_cast<u16*>(lds_build + 2u * kGrSlots + kGrSlots / 8u);  // [ref_stride] reference position -> node
  unsigned char* area = lds_build + 2u * kGrSlots + kGrSlots / 8u + ((2u * ws.ref_stride + 15u) & ~15u);
  u32* l_bits = reinterpret_cast<u32*>(area);                 // ranking: [nw] first instances of the survivors
  u32* l_base = l_bits + nw;                                  //          [kSeqCap] instance bases of the sequences
  u16* l_pref = reinterpret_cast<u16*>(l_base + kSeqCap);     //          [nw] survivors before the bitmap word
  u32* l_key = reinterpret_cast<u32*>(area);                  // edges:   [kGrSet] src << 15 | dst << 2 | kind
  u32* l_ok = l_key + kGrSet;                                 //          [kGrSet] order key of the first occurrence
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 const ns = seq_count(b, w);
  u32 const base_idx = b.read_win_off[w] + w;

  for (u32 i = threadIdx.x; i < kGrSlots / 2u; i += kGrT) reinterpret_cast<u32*>(l_node)[i] = 0xFFFFFFFFu;
  for (u32 i = threadIdx.x; i < kGrSlots / 32u; i += kGrT) l_sign[i] = 0;
  if (threadIdx.x == 0) l_fail = 0;
  IPROF_T0();
  // ---- 0. the general instances' read support (k_support's dedup mode queued one entry per count: table slot << 4 |
  //         sample << 1 | role): added up here as packed u16 (a k-mer has at most 2047 groups of mates) ----
  u32* const l_ctr = reinterpret_cast<u32*>(area);  // [tcap * CW / 2]
  u32 const n_cnt = (ws.mm_mode[a] & 0xE0000000u) == 0x20000000u ? ws.n_genq[a] : 0u;
  if (n_cnt) {
    u32 const ctr_words = (tcap * static_cast<u32>(CW) + 1u) / 2u;
    for (u32 i = threadIdx.x; i < ctr_words; i += kGrT) l_ctr[i] = 0;
    __syncthreads();
    const u32* genq = ws.slowq + static_cast<size_t>(a) * ws.inst_stride;
    constexpr int kQU = 4;
    for (u32 q0 = threadIdx.x; q0 < n_cnt; q0 += kGrT * kQU) {
      u32 e[kQU];
#pragma unroll
      for (int u = 0; u < kQU; ++u) e[u] = genq[min(q0 + u * kGrT, n_cnt - 1u)];
#pragma unroll
      for (int u = 0; u < kQU; ++u) {
        if (q0 + u * kGrT >= n_cnt) continue;
        u32 const sl = min(e[u] >> 4, tcap - 1u);
        u32 const i1 = sl * CW + ((e[u] >> 1) & 7u), i2 = sl * CW + S + (e[u] & 1u);
        atomicAdd(&l_ctr[i1 >> 1], 1u << ((i1 & 1u) * 16u));
        atomicAdd(&l_ctr[i2 >> 1], 1u << ((i2 & 1u) * 16u));
      }
    }
    __syncthreads();
  }
  IPROF(22);  // counts
  // ---- 1. survivors of RemoveLowCovNodes(0) (graph.cpp:363-390; node.cpp:38-42): their first instances into the bitmap ----
  constexpr u32 kPer = kGrSlots / kGrT;  // 6 slots per thread, every load of the trip in flight together
  u32 fi[kPer];
  u32 surv = 0;
#pragma unroll
  for (u32 j = 0; j < kPer; ++j) {
    u32 const s = threadIdx.x + j * kGrT;
    u32 const sc = s < tcap ? s : 0u;
    u64 const ky = keys[sc];
    fi[j] = first[sc];
    u32 total = 0;
    bool any = false, all = true, touched = false;
    for (int i = 0; i < CW; ++i) {
      u32 c = cnt[static_cast<size_t>(sc) * CW + i];
      if (n_cnt) {  // + the counts queued for this slot; the table row gets the sum (the node records and any general route read it)
        u32 const x = sc * CW + i;
        u32 const add = (l_ctr[x >> 1] >> ((x & 1u) * 16u)) & 0xFFFFu;
        if (add && s < tcap) {
          c += add;
          cnt[static_cast<size_t>(sc) * CW + i] = c;
          touched = true;
        }
      }
      if (i < S) {
        total += c;
        any |= c > 0;
        all &= c <= 1;
      }
    }
    (void)touched;
    bool const remove = (any && all) || total < min_node_cov;
    if (s < tcap && ky != 0 && !remove && fi[j] < ninst) surv |= 1u << j;
  }
  __threadfence_block();
  __syncthreads();  // the counters are dead: their area becomes the ranking area
  for (u32 i = threadIdx.x; i < nw; i += kGrT) l_bits[i] = 0;
  const u32* sbase = stage_seq_bases(ws.seq_inst_base + base_idx, ns, l_base);
  __syncthreads();
#pragma unroll
  for (u32 j = 0; j < kPer; ++j)
    if (surv & (1u << j)) atomicOr(&l_bits[fi[j] >> 5], 1u << (fi[j] & 31u));
  IPROF_SYNC(16);  // table pass
  // ---- 2. survivors before every bitmap word (block scan over the words' popcounts) ----
  u32 total_nodes;
  {
    u32 const per = (nw + kGrT - 1u) / kGrT;
    u32 const w0 = threadIdx.x * per;
    u32 mine = 0;
    for (u32 x = 0; x < per; ++x)
      if (w0 + x < nw) mine += __popc(l_bits[w0 + x]);
    u32 inc = mine;
    for (int d = 1; d < 64; d <<= 1) {
      u32 const y = __shfl_up(inc, d);
      if (lane >= d) inc += y;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    u32 before = 0, tot = 0;
    for (int x = 0; x < kGrT / 64; ++x) {
      u32 const t = sh[x];
      if (x < wave) before += t;
      tot += t;
    }
    total_nodes = tot;
    u32 run = before + inc - mine;
    if (tot <= kGrSlots)
      for (u32 x = 0; x < per; ++x)
        if (w0 + x < nw) {
          l_pref[w0 + x] = static_cast<u16>(run);
          run += __popc(l_bits[w0 + x]);
        }
  }
  IPROF_SYNC(17);  // ranks
  if (threadIdx.x == 0) ws.n_nodes[a] = total_nodes;
  if (total_nodes < ws.nc && total_nodes > kGrSlots) return;  // node indices beyond the packed edge key: the general kernels
  if (total_nodes >= ws.nc) {  // capacity exceeded: flagged (k_rank's rule), the retry passes grow it
    if (threadIdx.x == 0) {
      atomicOr(&ws.win_flags[w], 4u);
      ws.gr_done[a] = 1;
    }
    return;
  }
  // ---- 3. node records, a thread per survivor (k_rank 2b) ----
  u64 const win_read_off0 = b.read_off[b.read_win_off[w]];
#pragma unroll
  for (u32 j = 0; j < kPer; ++j) {
    if (!(surv & (1u << j))) continue;
    u32 const s = threadIdx.x + j * kGrT, ii = fi[j];
    u32 const idx = l_pref[ii >> 5] + __popc(l_bits[ii >> 5] & ((1u << (ii & 31u)) - 1u));
    l_node[s] = static_cast<u16>(idx);
    u32 const v = inst_slot[ii];
    u32 const sq = seq_of(sbase, ns, ii);
    u32 const o = ii - sbase[sq];
    u32 label = 1, srcbit = 0, rel_off = 0;  // Label::REFERENCE
    if (sq > 0) {
      u32 const r = b.read_win_off[w] + sq - 1;
      label = (b.read_flags[r] & MA_RF_CASE) ? 4u : 2u;  // Label::CASE / Label::CTRL
      srcbit = 0x80000000u;
      rel_off = static_cast<u32>(b.read_off[r] - win_read_off0);
    }
    for (int i = 0; i < S; ++i) ws.nd_cnt[(nb + idx) * S + i] = cnt[static_cast<size_t>(s) * CW + i];
    ws.nd_role[(nb + idx) * 2 + 0] = cnt[static_cast<size_t>(s) * CW + S];
    ws.nd_role[(nb + idx) * 2 + 1] = cnt[static_cast<size_t>(s) * CW + S + 1];
    ws.nd_src[nb + idx] = srcbit | (rel_off + o);
    ws.nd_label[nb + idx] = static_cast<u8>(label);
    ws.nd_sign[nb + idx] = (v & kInstPlus) ? 1 : 0;
    ws.nd_nedge[nb + idx] = 0;
    if (v & kInstPlus) atomicOr(&l_sign[idx >> 5], 1u << (idx & 31u));
  }
  IPROF_SYNC(18);  // node records
  // mRefNodeIds (graph.cpp:264-267): node of every reference k-mer, kNoNode when pruned
  SeqInfo const rsi = seq_info(b, w, 0, win_kmer(ws, w));
  u32* refn = ws.ref_node + static_cast<size_t>(a) * ws.ref_stride;
  for (u32 p = threadIdx.x; p < ws.ref_stride; p += kGrT) {
    u32 nd = kGrNone;
    if (p < rsi.nk) nd = l_node[min(ref_slot_g[p], kGrSlots - 1u)];
    l_refn[p] = static_cast<u16>(nd);
    refn[p] = nd == kGrNone ? kNoNode : nd;
  }
  for (u32 i = threadIdx.x; i < kGrSet; i += kGrT) {
    l_key[i] = kGrEmpty;
    l_ok[i] = 0xFFFFFFFFu;
  }
  IPROF_SYNC(19);  // reference nodes, set init
  // ---- 4. an edge and its mirror per (k+1)-mer whose two k-mers survive (graph.cpp:333-337), distinct ones into the set ----
  auto const h0_of = [&](u32 src) -> u32 { return __umulhi(src * 2654435761u, kGrSet); };
  auto const set_insert = [&](u32 src, u32 val, u32 okey) {
    u32 const key = (src << 15) | val;
    u32 h = h0_of(src);
    for (u32 probe = 0; probe < kGrProbe; ++probe) {
      u32 cur = l_key[h];
      if (cur == kGrEmpty) {
        u32 const old = atomicCAS(&l_key[h], kGrEmpty, key);
        cur = old == kGrEmpty ? key : old;
      }
      if (cur == key) {
        atomicMin(&l_ok[h], okey);
        return;
      }
      h = h + 1u == kGrSet ? 0u : h + 1u;
    }
    l_fail = 1;
  };
  auto const node_of = [&](u32 c) -> u32 {  // c: edge_word() of an instance word
    return (c >> 20) ? l_refn[min(c & 0xFFFFFu, ws.ref_stride - 1u)] : l_node[min(c & 0xFFFFFu, kGrSlots - 1u)];
  };
  auto const add_edge = [&](u32 ii, u32 ca, u32 cb) {
    u32 const na = node_of(ca), nbn = node_of(cb);
    if (na == kGrNone || nbn == kGrNone) return;
    // edge kind from the STORED signs of both nodes (graph.cpp:333-336)
    u32 const sa_minus = ((l_sign[na >> 5] >> (na & 31u)) & 1u) ^ 1u, sb_minus = ((l_sign[nbn >> 5] >> (nbn & 31u)) & 1u) ^ 1u;
    u32 const fwd = (sa_minus << 1) | sb_minus;                 // MakeFwdEdgeKind(sA, sB)
    u32 const rev = ((sb_minus ^ 1u) << 1) | (sa_minus ^ 1u);   // RevEdgeKind(fwd) seen from B
    set_insert(na, (nbn << 2) | fwd, 2u * ii);
    set_insert(nbn, (na << 2) | rev, 2u * ii + 1u);
  };
  // the reference's own (k+1)-mers (instance index == reference position), then the reads' from k_support's edge queue: the
  // ones that are not the reference's own edges again -- one in eight
  for (u32 p = threadIdx.x; p + 1 < nref; p += kGrT) add_edge(p, edge_word(inst_slot[p]), edge_word(inst_slot[p + 1]));
  {
    const uint2* edgeq = reinterpret_cast<const uint2*>(ws.slow_rec + static_cast<size_t>(a) * ws.inst_stride);
    u32 const neq = ws.n_edgeq[a];
    constexpr int kU = 4;
    for (u32 q0 = threadIdx.x; q0 < neq; q0 += kGrT * kU) {
      uint2 e[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) e[u] = edgeq[min(q0 + u * kGrT, neq - 1u)];
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (q0 + u * kGrT < neq) add_edge(e[u].x & 0x3FFFFFu, (e[u].x >> 22) | ((e[u].y & 0x7FFu) << 10), e[u].y >> 11);
    }
  }
  IPROF_SYNC(20);  // edge pass
  if (l_fail) return;  // (gr_done stays 0: the three general kernels redo this window from the table)
  // ---- 5. every distinct edge to its place in its node's list ----
  bool over = false;
  for (u32 i = threadIdx.x; i < kGrSet; i += kGrT) {
    u32 const key = l_key[i];
    if (key == kGrEmpty) continue;
    u32 const src = key >> 15, myok = l_ok[i];
    u32 r = 0, m = 0, h = h0_of(src);
    for (u32 probe = 0; probe < kGrSet; ++probe) {
      u32 const kk = l_key[h];
      if (kk == kGrEmpty) break;
      if ((kk >> 15) == src) {
        ++m;
        r += l_ok[h] < myok;
      }
      h = h + 1u == kGrSet ? 0u : h + 1u;
    }
    if (r < static_cast<u32>(kEdgeCap)) ws.nd_edge[(nb + src) * kEdgeCap + r] = key & 0x7FFFu;
    if (r == 0) ws.nd_nedge[nb + src] = static_cast<u8>(min(m, static_cast<u32>(kEdgeCap)));
    over |= m > static_cast<u32>(kEdgeCap);  // more than kEdgeCap distinct edges at one node
  }
  if (over) atomicOr(&ws.win_flags[w], 4u);
  if (threadIdx.x == 0) ws.gr_done[a] = 1;
  IPROF(21);  // edge lists
}

// ---- host side: one k attempt of the build stage for the active windows ----
int run_build_pass(ma_ctx* ctx, const DBatch& b, GraphWs& ws, u32* counters_dev, int tc_log2_alloc) {
  if (ws.n_active == 0) return MA_OK;
  int const S = ws.num_samples;
  // Round 5: NO host round trip inside a k attempt.  The table stride follows from the busiest window's slow instances
  // (counters_dev[0], k_classify) and is worked out by the kernels themselves (graph_ws.h: tbl_log2); the windows that need an
  // HBM-resident mate-mer set take theirs from the chunk's pool on the device (k_support).  The host used to read both
  // numbers back -- the stream drained twice per pass, per lane, per chunk, per rung of the ladder.
  size_t const A = ws.n_active;
  ws.tc_log2 = tc_log2_alloc;
  ws.max_slow = counters_dev;
  MA_HIP(ctx, hipMemsetAsync(counters_dev, 0, 12, ctx->stream));  // max_slow, max_gen[2]
  MA_HIP(ctx, hipMemsetAsync(ws.n_slow, 0, 4 * A, ctx->stream));
  MA_HIP(ctx, hipMemsetAsync(ws.mm_pool_used, 0, 8, ctx->stream));
  u32 const tiles_per_win = std::max<u32>(1, (ws.max_reads + 63) / 64);
  u32 const tile_cap = (64 * ws.max_read_len + 32 + 15) & ~15u;
  size_t const lds_c = ((ws.max_ref_len + 8 + 15) & ~15u) + 2048 + 2ull * tile_cap + 2 * 4ull * 64 * classify_mask_words(ws.max_read_len) + 64;
  if (lds_c > 160 * 1024) {
    ma_set_err(ctx, "ma_assemble_batch: reads too long for the LDS-staged classifier");
    return MA_ERR_PARAM;
  }
  if (lds_c > 65536)
    MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_classify), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds_c)));
  ctx->tic("k_classify");
  hipLaunchKernelGGL(k_classify, dim3(ws.n_active * tiles_per_win), dim3(kClsT), lds_c, ctx->stream, b, ws, counters_dev,
                     tiles_per_win, tile_cap);
  ctx->toc();
  // (k_insert initialises the slots each window uses: no table-wide memsets)
  ctx->tic("k_insert");
  hipLaunchKernelGGL(k_insert, dim3(ws.n_active), dim3(kInsT), 0, ctx->stream, b, ws);
  ctx->toc();
  // k_graph takes the common window (and then k_support queues that window's read-support counts for it)
  size_t const lds_g = graph_lds_bytes(ws.inst_stride, ws.ref_stride, S);
  ws.graph_fused = (lds_g <= 120u * 1024u && !getenv("MA_NO_GRAPH_FUSE")) ? 1u : 0u;
  // a wavefront's dedup table: the k-mers of two reads of the longest length at a load of 3/4 at most
  ws.dd_log2 = 9;
  while ((1u << ws.dd_log2) / 4u * 3u < 2u * ws.max_read_len && ws.dd_log2 < 12) ++ws.dd_log2;
  // + per-sequence cache (16 B per read of the busiest window, when that is at most kSupCache reads)
  u32 const sup_cache = ws.max_reads + 2 <= kSupCache ? ws.max_reads + 2 : 0u;
  u32 xs_log2 = 10;  // check (X)'s table: above the busiest window's read count (every read could bring a name of its own)
  while ((1u << xs_log2) < ws.max_reads + 2 && xs_log2 < 14) ++xs_log2;
  auto const sup_lds = [&](u32 lg) {
    return 2ull * ws.ref_stride * (S + 2) + 4 + (size_t(4 * (kSupT / 64)) << ws.dd_log2) + 4ull * (ws.max_reads + 2) + 9ull * (size_t(1) << lg) + 64 +
           16ull * sup_cache + 16;
  };
  while (xs_log2 > 10 && sup_lds(xs_log2) > 159u * 1024u) --xs_log2;  // (a table that fills up sends the window down the general route)
  size_t const lds_s = sup_lds(xs_log2);
  if (lds_s > 65536)
    MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_support), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds_s)));
  ctx->tic("k_support");
  ws.mm_probe_max = getenv("MA_MM_PROBE_MAX") ? static_cast<u32>(std::max(1, atoi(getenv("MA_MM_PROBE_MAX")))) : kMmLdsCap;
  hipLaunchKernelGGL(k_support, dim3(ws.n_active), dim3(kSupT), lds_s, ctx->stream, b, ws, counters_dev + 1, sup_cache, xs_log2);
  ctx->toc();
  // windows whose general instances fit an LDS set are finished by k_mm_lds; the HBM-resident sets only hold what the
  // remaining windows routed to them (usually nothing: both kernels then leave at their first test)
  ctx->tic("k_mm_q");
  hipLaunchKernelGGL(k_mm_q<14>, dim3(std::min<u32>(static_cast<u32>(ws.n_active), 512u)), dim3(kMmT), 0, ctx->stream, b, ws);
  ctx->toc();
  if (ws.max_reads + 1u > kSeqCap) {  // (the scan route is for windows of more sequences than a key's leader index names)
    ctx->tic("k_mm_lds");
    if (ws.mm_probe_max < kMmLdsCap) hipLaunchKernelGGL(k_mm_lds<true>, dim3(ws.n_active), dim3(kMmT), 0, ctx->stream, b, ws);
    else hipLaunchKernelGGL(k_mm_lds<false>, dim3(ws.n_active), dim3(kMmT), 0, ctx->stream, b, ws);
    ctx->toc();
  }
  ctx->tic("k_mm_hbm");
  hipLaunchKernelGGL(k_mm_hbm, dim3(ws.n_active), dim3(kBT), 0, ctx->stream, b, ws);
  ctx->toc();
  if (getenv("MA_VERBOSE")) {  // (diagnostics only: this does wait for the stream)
    std::vector<u32> mm(A);
    u32 max_gen[2] = {0, 0};
    unsigned long long used = 0;
    MA_HIP(ctx, ma_stream_sync(ctx));
    MA_HIP(ctx, hipMemcpy(mm.data(), ws.mm_mode, 4 * A, hipMemcpyDeviceToHost));
    MA_HIP(ctx, hipMemcpy(max_gen, counters_dev + 1, 8, hipMemcpyDeviceToHost));
    MA_HIP(ctx, hipMemcpy(&used, ws.mm_pool_used, 8, hipMemcpyDeviceToHost));
    size_t nfb = 0, big = 0;
    u64 tot = 0;
    for (u32 v : mm) {
      nfb += v >> 31;
      big += (v & 0x1FFFFFFFu) > kMmLdsMax;
      tot += v & 0x1FFFFFFFu;
    }
    fprintf(stderr, "[ma] mate-mer sets: %zu windows, %zu need the HBM set (%zu by size), mean general instances %.0f, max %u / %u; pool %.1f of %.1f MB\n",
            A, nfb, big, static_cast<double>(tot) / static_cast<double>(A), max_gen[0], max_gen[1], used / 1048576.0, ws.mm_pool_bytes / 1048576.0);
  }
  // node records + edges: k_graph for the common window (table of at most 8192 slots), the three general kernels for the rest
  MA_HIP(ctx, hipMemsetAsync(ws.gr_done, 0, 4 * A, ctx->stream));
  if (ws.graph_fused) {
    if (lds_g > 65536)
      MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_graph), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds_g)));
    ctx->tic("k_graph");
    hipLaunchKernelGGL(k_graph, dim3(ws.n_active), dim3(kGrT), lds_g, ctx->stream, b, ws, static_cast<u32>(ctx->prm.min_node_cov));
    ctx->toc();
  }
  ctx->tic("k_graph_gen");
  hipLaunchKernelGGL(k_graph_gen, dim3(std::min<u32>(static_cast<u32>(ws.n_active), 512u)), dim3(kRankT), 0, ctx->stream, b, ws,
                     static_cast<u32>(ctx->prm.min_node_cov));
  ctx->toc();
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

int run_count_inst(ma_ctx* ctx, const DBatch& b, const GraphWs& ws, int win0, int nwin, u32* maxima_dev) {
  MA_HIP(ctx, hipMemsetAsync(maxima_dev, 0, 32, ctx->stream));
  hipLaunchKernelGGL(k_count_inst, dim3(nwin), dim3(kBT), 0, ctx->stream, b, ws, win0, nwin, maxima_dev);
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

int run_select_active(ma_ctx* ctx, const GraphWs& ws, int win0, int nwin, const u32* gate_approx, u32* win_k,
                      u32* active, u32* n_active_dev) {
  MA_HIP(ctx, hipMemsetAsync(n_active_dev, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_select_active, dim3((nwin + 255) / 256), dim3(256), 0, ctx->stream, ws, win0, nwin,
                     gate_approx, win_k, active, n_active_dev, ctx->prm.min_k, ctx->prm.max_k, ctx->prm.k_step);
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

}  // namespace ma
