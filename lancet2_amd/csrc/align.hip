// Read <-> haplotype genotyping on gfx950: batched banded affine-gap alignment with in-kernel
// traceback, the CIGAR-walk allele-scoring epilogue, evidence de-duplication and site QUAL.
//
// Replaces caller::Genotyper::Genotype (caller/genotyper.cpp:224-235): AlignToAllHaplotypes
// (:376-411, minimap2 2.30 in the reference -- restated as the canonical seed-anchored overlap DP of
// the search region R = [vmin - K, vmax + K] is DERIVED from the seed diagonals, the
// read length and min_aln_score, there is no band parameter), AssignReadToAlleles (:269-362), ScoreReadAtVariant /
// ComputeLocalScore / ComputeSoftClipPenalty / ComputeEditDistance (caller/combined_scorer.cpp:24-108,
// caller/local_scorer.cpp:166-305, hts/cigar_utils.h:48-139), AddToTable + VariantSupport::AddEvidence
// (genotyper.cpp:423-456, caller/variant_support.cpp:24-30) and SomaticLogOddsRatio
// (caller/variant_call.cpp:316-345).
//
// Kernels:
//   k_read_planes   every read as three bit planes, once per batch
//   k_plan          per window: which haplotype slots get aligned, pair counts, the (window, haplotype) work list
//   k_plan_reads    wavefront per window: read -> window map, longest read, which samples are CASE samples
//   k_vote          workgroup per (window, haplotype): 11-mer chained index + haplotype bit planes in LDS; wave per
//                   read: shared 11-mers vote for their diagonal (unanimous votes skip the histogram); the extreme
//                   seed diagonals give the search region; the three gapless certificates settle most pairs right
//                   here; the others are classified by the width of their region
//   k_dp_scatter    DP list sorted by (width class, can the region reach a haplotype end)
//   k_align_reg<W>  one LANE per pair (inter-task SIMD, no cross-lane traffic); the (H,F) row of the region packed
//                   i16x2 in W+1 registers, row body fully unrolled (lean / general variant per wavefront and row),
//                   haplotype segment 4 bit/base in LDS, traceback nibbles written to HBM coalesced as
//                   [row][word][lane]; per-lane traceback -> CIGAR.  W = 33 .. 129 cells: a 150-base read needs 39 +
//                   the spread of its seed diagonals.  k_align_reg2: two classes of the same register budget side by side
//                   in one launch (a launch lasts its 150 dependent rows however few pairs it holds).
//   k_align_wave    one WAVEFRONT per pair for wider regions (seeds spread by tandem repeats / duplications): 64 cells
//                   of a row at a time, the horizontal gap chain closed with a wave prefix maximum, rows in LDS
//   k_align_gen     last resort for regions no LDS row holds: lane per pair, row in HBM
//   k_assign     one lane per read: best allele per variant over the haplotypes of each component
//   k_evidence   first read per (variant, sample, allele, qname) counts, by strand
//   k_qual       SOLOR site quality; germline PL / GQ / QUAL (variant-major: slot v of 64 consecutive windows per wavefront)
#include <algorithm>
#include <cstdlib>
#include <vector>

#include <type_traits>

#include "ma_internal.h"

// workgroups of the DP kernels in the order heaviest first (0: list order, as until round 6 -- developer A/B builds)
#ifndef MA_REG_ORDER
#define MA_REG_ORDER 1
#endif
namespace ma {

namespace {

constexpr int SK = 11;                 // seed length (minimap2 -k 11 in the reference)
constexpr int kIdxCap = 4096;          // hash buckets per haplotype index (>= 2 * max_hap_len rounded)
constexpr i32 GO = 12, GE = 3;         // scoring_constants.h:17-20
constexpr i32 NEGS = -16000;           // "minus infinity" that survives i16 packing
constexpr u32 kMinChainVotes = 4;      // exact 11-mers a chain of minimap2's min_chain_score 40 needs at least (oracle/align.cpp rule 2)
__constant__ u64 c_phred_bits_a[256] = {
#include "../../include/ma_phred_lut.inc"
};

__constant__ i8 c_score_matrix[25] = {1, -4, -4, -4, 0, -4, 1, -4, -4, 0, -4, -4, 1, -4, 0,
                                      -4, -4, -4, 1, 0, 0, 0, 0, 0, 0};  // scoring_constants.h:35-41

// Width classes of the DP kernels.  A pair whose region is wr diagonals wide runs in the first class that holds it.
constexpr int kNumReg = 6;
__host__ __device__ constexpr int reg_width(int c) {  // cells per row of k_align_reg's instantiations
  return c == 0 ? 33 : c == 1 ? 41 : c == 2 ? 49 : c == 3 ? 65 : c == 4 ? 97 : 129;
}
// wider regions: one WAVEFRONT per pair (k_align_wave), two LDS footprints; the lane-per-pair kernel with its row in HBM
// (k_align_gen) only for regions no LDS row holds
constexpr int kClsWaveS = kNumReg, kClsWaveB = kNumReg + 1, kClsGlobal = kNumReg + 2, kNumCls = kNumReg + 3;
constexpr int kNumKeys = 2 * kNumCls;  // key = class * 2 + (region can reach a haplotype end)
constexpr u32 kWaveSmallW = 0;         // (one wavefront class: a launch costs a fixed ~0.2 ms of latency, its LDS is sized for the
                                       //  widest region that actually occurs)

struct AlnWs {
  u32 wave_big_w;      // widest region the big wavefront class holds for this batch's longest read (host computed)
  // planning
  u32* win_slotmask;   // [n] bitmask of haplotype slots to align
  u32* win_case;       // [n] bitmask of the samples that have a CASE read in the window (k_plan_reads)
  u64* pair_off;       // [n + 1]
  u32* counters;       // [8]: 0 max read len, 1 max reads per window, 2 entries in vote_wg
  u32* vote_wg;        // [n * MH] compact list of (window * MH + slot) to align against
  u32* read_win;       // [n_reads] window of every read (k_read_planes writes it: no binary search per lane later)
  u32* read_planes;    // [n_reads][3][rwords] every read as three bit planes (k_read_planes), read by each haplotype's k_vote
  // haplotype seed index, per (window, slot)
  // per pair
  i32* centre;         // [pairs in chunk] first diagonal of the pair's region (vmin - K), or a sentinel
  u32* band_w;         // [pairs in chunk] width of the region in diagonals | key << 16
  u32* pair_read;      // [pairs in chunk] DP pairs only: read index | haplotype slot << 27 (no binary search over pair_off later)
  u64* vote_aux;       // [pairs in chunk] wide pairs only: most-voted diagonal - region start | votes further than 8 / 16 / 24
                       //                  diagonals from it (saturating u16 each): k_align_wave's narrow first pass
  u32* dp_list;        // [pairs in chunk] pairs that need the DP (compacted by k_vote)
  u32* dp_count;       // device counters: [0] DP pairs, [4 ..] pairs per key, [40] / [41] widest region of the HBM / wavefront class
  u32* tb;             // traceback nibbles
  u32 tb_words;        // words per row (of the launch)
  u32 tb_rows;         // rows per pair (max read len + 1)
  u32* gen_row;        // k_align_gen's (H,F) rows (HBM)
  u32 gen_w;           // widest region of this k_align_gen / k_align_wave launch
  unsigned long long* tbw;  // k_align_wave's traceback bit planes
  // evidence table per window
  u32 ev_cap;
  u64* ev_key;         // [n][ev_cap]
  u32* ev_min;         // [n][ev_cap]
  u8* asg_allele;      // [n_reads * MV] (internal copy when the caller passes NULL)
  // alignment records, COMPACT: one per planned (read, haplotype) pair, indexed by the global pair index
  // pair_off[w] + (rank of the haplotype slot among the window's aligned slots) * reads of the window + read -- the reads of
  // one (window, haplotype) are consecutive, so a wavefront of k_assign (a lane per read) loads 64 records as one run.
  // (Round 4 kept them in the caller's fixed-stride layout [read][max_haps]: 1.5 KB between two lanes' records, 8 GB of
  //  workspace touched a sector at a time.)  The caller's debug taps aln_rec / aln_cigar are filled from these (k_tap_records).
  i32* rec;            // [pairs][8]  packed record (rec_store)
  u32* cig;            // [pairs][1 + max_cigar]  op count, ops -- written for DP pairs only, read when a CIGAR has more than four operations
};

struct GArgs {
  DBatch b;
  ma_asm_out_t a;
  ma_var_out_t v;
  ma_geno_out_t o;
  AlnWs ws;
  ma_params_t prm;
  u64 pair0;     // first global pair index of this chunk
  u32 npairs;    // pairs in this chunk
  u32 dp0;       // first dp_list entry of this DP launch
  u32 dp_n;      // entries of this DP launch
};

// A pair's record is 32 bytes: [0] score, [1] rs | re << 16, [2] qs | qe << 16 (haplotype and read coordinates are below 2^16),
// [3] number of CIGAR operations (0: no alignment), [4..7] the first four operations -- a certificate pair is `S? M S?`, and 94 %
// of the pairs are certificate pairs.  Longer CIGARs (DP pairs) are ALSO in the side array cig[pair][1 + max_cigar]; nobody
// touches a pair's side slot otherwise (round 4 wrote 24 + 8 bytes into a 24-byte record and a 68-byte slot per pair).
constexpr u32 kRecWords = 8;
__device__ __forceinline__ u32* rec_at(AlnWs const& ws, u64 gp) { return reinterpret_cast<u32*>(ws.rec) + gp * kRecWords; }
__device__ __forceinline__ u32* cig_at(AlnWs const& ws, ma_params_t const& prm, u64 gp) { return ws.cig + gp * (1 + prm.max_cigar); }
__device__ __forceinline__ void rec_store(u32* r, i32 score, i32 rs, i32 re, i32 qs, i32 qe, u32 nops, u32 o0, u32 o1, u32 o2, u32 o3) {
  reinterpret_cast<uint4*>(r)[0] = make_uint4(static_cast<u32>(score), static_cast<u32>(rs) | (static_cast<u32>(re) << 16),
                                              static_cast<u32>(qs) | (static_cast<u32>(qe) << 16), nops);
  reinterpret_cast<uint4*>(r)[1] = make_uint4(o0, o1, o2, o3);
}

// A read's three bit planes live in one wavefront (k_vote / k_align_*: a lane per word): 608 bases at most.  A window
// that holds a longer read is not genotyped and says so (MA_W_READ_OVERFLOW) -- the batch goes on without it.
constexpr u32 kMaxGenoRead = 608;
__global__ __launch_bounds__(256) void k_max_reads(DBatch b, u32* win_status, u32* out) {  // out[0] most reads of a window, out[1] longest read
  // one wavefront per window, a lane per read
  int const w = blockIdx.x * 4 + static_cast<int>(threadIdx.x >> 6);
  if (w >= b.n_windows) return;
  u32 const lane = threadIdx.x & 63u;
  u32 ml = 0;
  for (u32 r = b.read_win_off[w] + lane; r < b.read_win_off[w + 1]; r += 64) ml = max(ml, static_cast<u32>(b.read_off[r + 1] - b.read_off[r]));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) ml = max(ml, static_cast<u32>(__shfl_xor(ml, d, 64)));
  if (lane != 0) return;
  atomic_max_lazy(out, b.read_win_off[w + 1] - b.read_win_off[w]);
  // (a window the assembler skipped for its reads keeps that flag: it has no haplotypes to genotype)
  u32 const st = win_status[w] & ~static_cast<u32>(MA_W_CIGAR_OVERFLOW);
  if (ml > kMaxGenoRead) {
    win_status[w] = st | static_cast<u32>(MA_W_READ_OVERFLOW);
  } else {
    win_status[w] = (st & MA_W_NO_HAPLOTYPE) ? st : (st & ~static_cast<u32>(MA_W_READ_OVERFLOW));
    atomic_max_lazy(out + 1, ml);
  }
}

// ---- planning ----
__global__ void k_plan(GArgs A) {
  int const w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= A.b.n_windows) return;
  ma_params_t const& P = A.prm;
  u32 mask = 0, hl = 0;
  u32 const nv = A.v.win_nvars[w];
  if (nv > 0 && !(A.a.win_status[w] & (MA_W_NO_HAPLOTYPE | MA_W_READ_OVERFLOW))) {
    for (u32 c = 0; c < A.a.win_ncomp[w]; ++c) {
      bool has = false;
      for (u32 x = 0; x < nv && !has; ++x) has = A.v.var_comp[static_cast<size_t>(w) * P.max_vars + x] == c;
      if (!has) continue;  // variant_builder.cpp:248: components without variants are not genotyped
      size_t const ci = static_cast<size_t>(w) * P.max_comps + c;
      for (u32 h = 0; h < A.a.comp_nhaps[ci]; ++h) {
        mask |= 1u << (A.a.comp_hap0[ci] + h);
        hl = max(hl, A.a.hap_len[static_cast<size_t>(w) * P.max_haps + A.a.comp_hap0[ci] + h]);
      }
    }
  }
  if (hl) atomic_max_lazy(&A.ws.counters[3], hl);  // longest haplotype that is aligned: sizes k_vote's LDS
  A.ws.win_slotmask[w] = mask;
  if (mask) {  // dense work list: a (window, slot) grid would leave most workgroups (and whole XCDs) empty
    u32 at = atomicAdd(&A.ws.counters[2], static_cast<u32>(__popc(mask)));
    for (u32 mm = mask; mm; mm &= mm - 1) A.ws.vote_wg[at++] = static_cast<u32>(w) * P.max_haps + (__ffs(mm) - 1);
  }
  u32 const nr = A.b.read_win_off[w + 1] - A.b.read_win_off[w];
  A.ws.pair_off[w] = static_cast<u64>(nr) * __popc(mask);  // counts; scanned below
  atomic_max_lazy(&A.ws.counters[1], nr);
}

// The per-read part of the plan, one wavefront per window (a thread per window walking its 600 reads was 600 dependent
// iterations): read -> window map, longest read, and which samples are CASE samples (k_qual asked every variant's thread
// to find that out from the window's reads again).
__global__ __launch_bounds__(256) void k_plan_reads(GArgs A) {
  int const w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= A.b.n_windows) return;
  u32 const r0 = A.b.read_win_off[w], r1 = A.b.read_win_off[w + 1];
  u32 const S = static_cast<u32>(A.prm.num_samples);
  u32 ml = 0, cm = 0;
  for (u32 r = r0 + lane; r < r1; r += 64) {
    ml = max(ml, static_cast<u32>(A.b.read_off[r + 1] - A.b.read_off[r]));
    A.ws.read_win[r] = static_cast<u32>(w);
    u32 const smp = A.b.read_sample[r];
    if ((A.b.read_flags[r] & MA_RF_CASE) && smp < S) cm |= 1u << smp;
  }
  for (int off = 32; off > 0; off >>= 1) {
    ml = max(ml, static_cast<u32>(__shfl_xor(ml, off)));
    cm |= static_cast<u32>(__shfl_xor(cm, off));
  }
  if (lane == 0) {
    A.ws.win_case[w] = cm;
    if (ml && !(A.a.win_status[w] & MA_W_READ_OVERFLOW)) atomic_max_lazy(&A.ws.counters[0], ml);
  }
}

__global__ void k_scan_pairs(u64* pair_off, int n) {  // single block exclusive scan (n <= few 100k)
  __shared__ u64 carry;
  __shared__ u64 sh[1024];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int const i = base + threadIdx.x;
    u64 const v = i < n ? pair_off[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      u64 const x = threadIdx.x >= static_cast<u32>(d) ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += x;
      __syncthreads();
    }
    u64 const incl = sh[threadIdx.x];
    if (i < n) pair_off[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) pair_off[n] = carry;
}

// pair p -> (window, read, slot)
struct PairId { int w; u32 r; u32 slot; };
__device__ PairId pair_decode(GArgs const& A, u64 p) {
  int lo = 0, hi = A.b.n_windows;  // largest w with pair_off[w] <= p
  while (hi - lo > 1) {
    int const mid = (lo + hi) / 2;
    if (A.ws.pair_off[mid] <= p) lo = mid; else hi = mid;
  }
  PairId id;
  id.w = lo;
  u64 const local = p - A.ws.pair_off[lo];
  u32 const nr = A.b.read_win_off[lo + 1] - A.b.read_win_off[lo];
  u32 const si = static_cast<u32>(local / nr);
  id.r = A.b.read_win_off[lo] + static_cast<u32>(local % nr);
  u32 mask = A.ws.win_slotmask[lo];
  for (u32 x = 0; x < si; ++x) mask &= mask - 1;
  id.slot = __ffs(mask) - 1;
  return id;
}

#if defined(MA_PROFILE) || defined(MA_PROFILE_TRIPS)
__device__ unsigned long long g_vprof[16];
#endif
#ifdef MA_PROFILE
#define VPROF_T0() unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define VPROF_ACC(slot)                                                       \
  do {                                                                        \
    unsigned long long _t1 = __builtin_amdgcn_s_memtime();                    \
    if (lane == 0) ix.prof[slot] += _t1 - _t0;                                \
    _t0 = _t1;                                                                \
  } while (0)
#else
#define VPROF_T0() do {} while (0)
#define VPROF_ACC(slot) do {} while (0)
#endif

// ---- every read as three bit planes, once (each of the window's haplotypes votes against the same planes) ----
// plane 0 / 1: low / high bit of the 2-bit base code, plane 2: base is not A/C/G/T; bit b of word b >> 5; zero padded
// words per read record: three planes of rwords words, padded to whole 128-byte lines (full-line stores, one or two
// aligned lines per fetch)
__host__ __device__ constexpr u32 plane_stride(u32 rwords) { return (3u * rwords + 31u) & ~31u; }
// A lane per (read, 32-base word): 32 bytes in as dwords (any alignment), three plane words out.  G = 8 / 16 / 32 lanes per
// read (the power of two that holds rwords); four ASCII bases become four 2-bit codes per dword operation (swar below), an
// all-A/C/G/T dword -- all but one in a few hundred -- needs nothing else, anything else is encoded byte by byte with
// enc_base.  (Round 5; the first form packed one read per wavefront and trip -- a byte per lane, nine ballots and an LDS
// round per 64 bases, 17 reads in series: 2.2 ms per 16384 windows for 2.6 GB of traffic.)
__device__ __forceinline__ u32 nib4(u32 c) {  // bit 0 of each of the four bytes -> bits 0..3
  u32 const y = (c | (c >> 7)) & 0x00030003u;
  return (y | (y >> 14)) & 0xFu;
}
template <u32 G>
__global__ __launch_bounds__(256) void k_read_planes(GArgs A, u32 rwords) {
  i64 const r = static_cast<i64>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  u32 const x = threadIdx.x % G;
  if (r >= A.b.n_reads || x >= rwords) return;
  u64 const ro = A.b.read_off[r];
  i32 m = static_cast<i32>(A.b.read_off[r + 1] - ro);
  if (m > static_cast<i32>(rwords - 2) * 32) m = 0;  // a read of a window that is not genotyped (MA_W_READ_OVERFLOW)
  i32 const nb = min(max(m - 32 * static_cast<i32>(x), 0), 32);
  u32 lo = 0, hi = 0, bad = 0;
  if (nb > 0) {
    const u8* rb = A.b.read_bases + ro + 32u * x;
    u32 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      i32 const rem = nb - 4 * q;
      v[q] = 0x41414141u;  // (past the read's end: 'A' -- code 0, nothing flagged)
      if (rem >= 4) {
        __builtin_memcpy(&v[q], rb + 4 * q, 4);
      } else if (rem > 0) {  // the read's last bytes: nothing behind them is touched (the caller's buffer may end there)
        u32 t = 0x41414141u;
        for (i32 y = 0; y < rem; ++y) t = (t & ~(0xFFu << (8 * y))) | (static_cast<u32>(rb[4 * q + y]) << (8 * y));
        v[q] = t;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      u32 c = (v[q] >> 1) & 0x03030303u;  // A 0, C 1, T 2, G 3
      c ^= (c >> 1) & 0x01010101u;        // A 0, C 1, G 2, T 3
      u32 const c0 = c & 0x01010101u, c1 = (c >> 1) & 0x01010101u;
      // the byte a code stands for: 0x41 ^ 0x02 c0 ^ 0x06 c1 ^ 0x11 c0 c1 (A 0x41, C 0x43, G 0x47, T 0x54), shifts and xors only
      u32 const both = c0 & c1;
      u32 const expect = 0x41414141u ^ (c0 << 1) ^ (c1 << 1) ^ (c1 << 2) ^ both ^ (both << 4);
      u32 l4 = nib4(c0), h4 = nib4(c1), b4 = 0;
      if (v[q] != expect) {  // lower case, N, anything else: enc_base's rule, byte by byte
        l4 = h4 = 0;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          u32 const e = enc_base(static_cast<u8>((v[q] >> (8 * y)) & 0xFFu));
          l4 |= (e & 1u) << y;
          h4 |= ((e >> 1) & 1u) << y;
          b4 |= (e > 3u ? 1u : 0u) << y;
        }
      }
      lo |= l4 << (4 * q);
      hi |= h4 << (4 * q);
      bad |= b4 << (4 * q);
    }
  }
  u32* out = A.ws.read_planes + static_cast<size_t>(r) * plane_stride(rwords);
  out[x] = lo;
  out[rwords + x] = hi;
  out[2 * rwords + x] = bad;
  for (u32 y = 3 * rwords + x; y < plane_stride(rwords); y += rwords) out[y] = 0u;  // (the stride's padding reads as zero)
}

// ---- seed vote: one workgroup per (window, haplotype), one wave per read ----
// The haplotype's 11-mer index (bucket heads + chains + codes) lives in LDS for the lifetime of the
// workgroup, so the per-read work never leaves the CU: every read of the window votes against it.
struct HapIdx {
  const u16* head;  // [kIdxCap]      0xFFFF = empty
  const u16* next;  // [max_hap_len]
  const u32* code;  // [max_hap_len]  0xFFFFFFFF = no valid 11-mer
  const u32* hlo;   // bit planes of the haplotype's 2-bit base codes: bit j of word j >> 5
  const u32* hhi;
  const u32* hbad;  // base j is not A/C/G/T
  u32* rplanes;     // this wave's read planes: [3][rwords]
  u32 rwords;
  u32* dpbuf;       // this wave's pending DP pairs: [64] + count at [64] + their keys at [65 .. 129)
  u32 hap_amb;      // the haplotype holds a base that is not A/C/G/T
  const u16* dup_pre;  // [n + 1] sum of (occurrences - 1) of the 11-mers starting before position j (k_vote's set-up)
  const i32* cand;     // hint shortcut: [0] number of candidate shifts (0: off), [1] anchor, [2 ..] shifts
#ifdef MA_PROFILE
  unsigned long long* prof;  // this wave's phase cycle counters
#endif
};
// 11 consecutive bits of a bit plane starting at bit i (planes are padded with two zero words)
__device__ __forceinline__ u32 plane11(const u32* pl, u32 i) {
  u64 const two = static_cast<u64>(pl[i >> 5]) | (static_cast<u64>(pl[(i >> 5) + 1]) << 32);
  return static_cast<u32>(two >> (i & 31u)) & 0x7FFu;
}
// 32 consecutive bits of a bit plane starting at bit i
__device__ __forceinline__ u32 plane32(const u32* pl, u32 i) {
  u64 const two = static_cast<u64>(pl[i >> 5]) | (static_cast<u64>(pl[(i >> 5) + 1]) << 32);
  return static_cast<u32>(two >> (i & 31u));
}
// index entry == read code?  Bit 31 of an entry flags "this 11-mer occurs more than once in the haplotype".
constexpr u32 kCodeDup = 0x80000000u;
__device__ __forceinline__ bool same_code(u32 entry, u32 cd) { return ((entry ^ cd) & ~kCodeDup) == 0; }
// All-reduce over the eight lanes 8 g .. 8 g + 7 of a wave (k_vote: one read per group) on the DPP path: lane ^ 1 and
// lane ^ 2 inside the quad, then the mirror image inside the half row (lane i <-> 7 - i: the other quad).  No LDS traffic --
// the kernel is bound by the CU's LDS pipe and __shfl_xor is a ds_bpermute.  (All 64 lanes must be active.)
template <class Op>
__device__ __forceinline__ u32 grp8_reduce(u32 v, Op op) {
  v = op(v, static_cast<u32>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xF, 0xF, false)));   // quad_perm [1,0,3,2]
  v = op(v, static_cast<u32>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x4E, 0xF, 0xF, false)));   // quad_perm [2,3,0,1]
  v = op(v, static_cast<u32>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x141, 0xF, 0xF, false)));  // row_half_mirror
  return v;
}
__device__ __forceinline__ u32 grp8_sum(u32 v) { return grp8_reduce(v, [](u32 a, u32 b) { return a + b; }); }
__device__ __forceinline__ u32 grp8_or(u32 v) { return grp8_reduce(v, [](u32 a, u32 b) { return a | b; }); }
__device__ __forceinline__ i32 grp8_min(i32 v) {
  return static_cast<i32>(grp8_reduce(static_cast<u32>(v), [](u32 a, u32 b) { return static_cast<u32>(min(static_cast<i32>(a), static_cast<i32>(b))); }));
}
__device__ __forceinline__ i32 grp8_max(i32 v) {
  return static_cast<i32>(grp8_reduce(static_cast<u32>(v), [](u32 a, u32 b) { return static_cast<u32>(max(static_cast<i32>(a), static_cast<i32>(b))); }));
}
// append this wave's buffered DP pairs to the global list: one atomic per 64 pairs
__device__ __forceinline__ void vote_flush_dp(GArgs const& A, HapIdx ix, int lane) {
  u32 const cnt = ix.dpbuf[64];
  if (cnt == 0) return;
  u32 base = 0;
  if (lane == 0) base = atomicAdd(A.ws.dp_count, cnt);
  base = __shfl(base, 0);
  if (static_cast<u32>(lane) < cnt) A.ws.dp_list[base + lane] = ix.dpbuf[lane];
  u32 const key = static_cast<u32>(lane) < cnt ? ix.dpbuf[65 + lane] : 0xFFFFFFFFu;
  for (unsigned long long todo = __ballot(static_cast<u32>(lane) < cnt); todo;) {  // pairs per key: one atomic per key present
    u32 const k0 = __shfl(key, __builtin_ctzll(todo));
    unsigned long long const same = __ballot(key == k0);
    if (lane == 0) atomicAdd(&A.ws.dp_count[4 + k0], static_cast<u32>(__popcll(same)));
    todo &= ~same;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) ix.dpbuf[64] = 0;
  __builtin_amdgcn_wave_barrier();
}
// "no alignment": the hit flag of the pair's record is cleared explicitly (every reader tests it before anything else),
// so the multi-GB internal record arrays need no memset per batch.
__device__ __forceinline__ void write_no_hit(GArgs const& A, u64 lp) {
  rec_at(A.ws, A.pair0 + lp)[3] = 0;
}

// What the votes of one pair come to, decided by ONE lane (vote_pair's lane 0; the first lane of a read's group in k_vote's
// eight-reads-per-trip vote): the gapless certificates (I) / (II) write the record, (III) writes "no alignment", everything
// else is prepared for the DP kernels -- the return value is then the pair's (width class, wall) key, for the caller to queue.
// c = the most-voted diagonal, [vmin, vmax] the extreme anchor diagonals, K the reach, X / amb the mismatches / ambiguous
// bases of the gapless path on c, v2 the runner-up's votes, v_off the votes not on c.  (The reasoning is with vote_pair.)
constexpr u32 kVoteFast = 0xFFFFFFFEu, kVoteNoHit = 0xFFFFFFFFu;
__device__ __forceinline__ u32 vote_settle(GArgs const& A, u64 lp, PairId id, i32 m, i32 n, i32 c, i32 vmin, i32 vmax, i32 K, i32 X, bool amb,
                                           bool ramb, bool hap_amb, i32 v2, i32 v_off, u32 vfar8, u32 vfar16, u32 vfar24,
                                           i32 lb_other = -(1 << 20), i32 n_amb = 0) {
  i32 const ms = A.prm.min_aln_score;
  i32 const o_left = c < 0 ? -c : 0, o_right = c + m > n ? c + m - n : 0;
  bool const inside = o_left == 0 && o_right == 0;
  i32 const qs = o_left, qe = m - o_right;  // overlap rows [qs, qe)
  i32 const S0 = (qe - qs) - 5 * X;
  bool const certs = !(A.prm.aln_tier & 2);
  bool const ok_common = certs && !amb && X <= 2 && S0 >= ms && m < (1 << 27);
  bool const fast_in = inside && ok_common && v2 + 10 + 11 * X < m;
  bool const fast_ov = !inside && (o_left == 0 || o_right == 0) && ok_common && !ramb && !hap_amb &&
                       11 * S0 > 6 * m + 50 + 5 * v_off && A.prm.max_cigar >= 2;
  i32 const L0 = qe - qs, lmax = min(m, L0 + K + (o_left ? vmax - c : c - vmin));
  bool const nohit = certs && !inside && (o_left == 0 || o_right == 0) && !ramb && !hap_amb && S0 < ms && L0 - 14 < ms &&
                     6 * lmax + 50 + 5 * v_off < 11 * ms;
  if (nohit) {
    A.ws.centre[lp] = 0x7FFFFFFF;  // no alignment
    write_no_hit(A, lp);
    return kVoteNoHit;
  }
  if (fast_in || fast_ov) {
    // BuildCigar (genotyper.cpp:45-69): S(qs) + core + S(m - qe); ops: 0 M, 4 S
    u32 o[3] = {0u, 0u, 0u};
    u32 nc = 0;
    u32 const ops_s = (static_cast<u32>(qs) << 4) | 4u, ops_m = static_cast<u32>(qe - qs) << 4, ops_e = (static_cast<u32>(m - qe) << 4) | 4u;
    if (qs > 0) {
      o[0] = ops_s;
      o[1] = ops_m;
      nc = 2;
    } else {
      o[0] = ops_m;
      nc = 1;
    }
    if (qe < m) {
      if (nc == 1) o[1] = ops_e; else o[2] = ops_e;
      ++nc;
    }
    rec_store(rec_at(A.ws, A.pair0 + lp), S0, c + qs, c + qe, qs, qe, nc, o[0], o[1], o[2], 0u);
    A.ws.centre[lp] = 0x7FFFFFFE;
    return kVoteFast;
  }
  // ---- the DP's region, narrowed EXACTLY (round 5) --------------------------------------------------------------------
  // Rule 3 sizes R = [vmin - K, vmax + K] for the weakest alignment that counts (score min_aln_score).  The DP's answer --
  // the optimum inside R with the canonical ties -- is known to score at least S_lb whenever some overlap alignment inside
  // [vmin, vmax] is: the gapless path P0 on c (S0), or the caller's one-gap path between two voted diagonals (lb_other).
  // With nothing ambiguous in read or haplotype:
  //   (a) a path of score >= S_lb has gaps of total length <= K' = (m - S_lb - 12) / 3: it stays within K' diagonals of
  //       any diagonal it visits;
  //   (b) with L paired rows, x mismatches and g gaps it scores <= L - 5 x - 15 g and holds >= L - 10 (g + 1) - 11 x >=
  //       S_lb - 10 - 6 (m - S_lb) / 5 exact 11-mers -- at least kMinChainVotes of them once 11 S_lb >= 70 + 6 m -- all on its
  //       own diagonals, which lie within K' <= K of each other: every voted diagonal of the path is an ANCHOR, so the path
  //       visits [vmin, vmax] and lies inside R' = [vmin - K', vmax + K'];
  //   (c) every optimal path inside R, and every optimal-or-tied prefix of one (prefix + the path's suffix is such a path),
  //       therefore lies inside R': the cells the end-cell choice and the traceback compare hold the same H / E / F in the
  //       DP over R' as over R, every other cell at most what it held -- the same end cell, the same moves, the same record.
  // The kernels' cost is rows x the width class of the region: 39 + spread columns become 2 K' + 1 + spread.
  //   (d) rows that NO path inside the region can pair are paid by every path: a region whose largest diagonal is D < 0
  //       clips >= -D rows at the read's start, one that ends past the haplotype's end likewise at its end.  They come off
  //       the rows a path can pair in (b) -- with the full K, before anything is narrowed -- and off the gap budget in (a),
  //       which may then be re-applied to the narrower region: a read hanging 65 bases over a haplotype end with one
  //       mismatch (11 S0 <= 6 m + 50, no certificate) has K' = 17 by (a) alone and 0 with the 64 rows it cannot pair.
  i32 Kn = K;
  {
    i32 s_lb = lb_other;
    // (an N in the READ -- n_amb of them, none in the haplotype: X counts it as a mismatch, so S0 only underestimates P0; a
    //  path pairs at most n_amb of them at cost 2 each, and each spoils 11 of its 11-mers: 45 n_amb more in (b)'s condition)
    if ((o_left == 0 || o_right == 0) && qe - qs > 0 && c >= vmin && c <= vmax) s_lb = max(s_lb, S0);
    i32 const clip0 = max(0, -(vmax + K)) + max(0, vmin - K + m - n);
    if (certs && !hap_amb && (ramb || !amb) && 11 * s_lb >= 70 + 6 * (m - clip0) + 45 * n_amb) {
      Kn = min(K, max(0, (m - s_lb - GO) / GE));
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        i32 const clip = max(0, -(vmax + Kn)) + max(0, vmin - Kn + m - n);
        Kn = min(Kn, max(0, (m - s_lb - clip - GO) / GE));
      }
    }
  }
#ifdef MA_DP_HIST  // (developer build: how wide are the DP regions?  dp_count[24 ..])
  if (!(fast_in || fast_ov || nohit)) {
    u32 const w = static_cast<u32>(vmax - vmin + 2 * Kn + 1);
    atomicAdd(&A.ws.dp_count[24 + (w <= 5 ? 0 : w <= 9 ? 1 : w <= 13 ? 2 : w <= 17 ? 3 : w <= 25 ? 4 : w <= 33 ? 5 : w <= 41 ? 6 : 7)], 1u);
  }
#endif
  i32 const r_lo = vmin - Kn;
  u32 const r_w = static_cast<u32>(vmax - vmin + 2 * Kn + 1);
  // width class of the region; can a row of the kernel's window reach column 0 or n (+ the 7 columns the last
  // segment word carries beyond it)?  k_align_reg picks its row body per wavefront, so the two kinds are kept apart
  int cls = kNumReg;
  if (!(A.prm.aln_tier & 1))
    for (cls = 0; cls < kNumReg && r_w > static_cast<u32>(reg_width(cls)); ++cls) {}
  if (cls == kNumReg) cls = r_w <= kWaveSmallW ? kClsWaveS : (r_w <= A.ws.wave_big_w ? kClsWaveB : kClsGlobal);
  i32 const kw = cls < kNumReg ? reg_width(cls) : static_cast<i32>(r_w);
  u32 const wall = (r_lo >= 0 && r_lo + kw + m + 8 <= n) ? 0u : 1u;
  u32 const key = static_cast<u32>(cls) * 2u + wall;
  if (cls == kClsGlobal) atomic_max_lazy(&A.ws.dp_count[40], r_w);
  if (cls == kClsWaveB) atomic_max_lazy(&A.ws.dp_count[41], r_w);
  if (cls >= kNumReg)
    A.ws.vote_aux[lp] = static_cast<u64>(static_cast<u32>(c - r_lo) & 0xFFFFu) | (static_cast<u64>(min(vfar8, 65535u)) << 16) |
                        (static_cast<u64>(min(vfar16, 65535u)) << 32) | (static_cast<u64>(min(vfar24, 65535u)) << 48);
  A.ws.centre[lp] = r_lo;
  A.ws.band_w[lp] = r_w | (key << 16);
  A.ws.pair_read[lp] = id.r | (id.slot << 27);
  return key;
}

__device__ __forceinline__ void vote_pair(GArgs const& A, u64 lp, PairId id, HapIdx ix, u16* hist, int lane, i32 m, u32 pre, i32 hint, i32 n,
                                          bool shortcut_tried);

#ifndef MA_VOTE_PF
#define MA_VOTE_PF 2
#endif
// waves per SIMD the registers are held to (96 VGPRs): five workgroups per CU, as many as their LDS allows -- the
// kernel spends most of its life waiting for LDS and memory, one more resident workgroup in four is worth 13 % (a sixth,
// bought with a half-size seed index, costs more in longer bucket chains than it hides)
#ifndef MA_VOTE_WAVES
#define MA_VOTE_WAVES 5
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MA_VOTE_WAVES, 8))) void k_vote(GArgs A, u32 hist_len, u32 rwords, u32 ml_eff) {
  extern __shared__ u32 lds_vote[];
#if defined(MA_PROFILE) || defined(MA_PROFILE_TRIPS)
  unsigned long long const k_tstart = __builtin_amdgcn_s_memtime();
#endif
  u32 const item = A.ws.vote_wg[blockIdx.x];
  int const w = item / A.prm.max_haps, slot = item % A.prm.max_haps;
  u32 const mask = A.ws.win_slotmask[w];
  u32 const si = __popc(mask & ((1u << slot) - 1u));
  u32 const r0 = A.b.read_win_off[w], nr = A.b.read_win_off[w + 1] - r0;
  u64 const p0 = A.ws.pair_off[w] + static_cast<u64>(si) * nr;  // global pair index of read 0
  if (p0 + nr <= A.pair0 || p0 >= A.pair0 + A.npairs) return;
  // LDS is carved for the longest haplotype that is actually aligned in this batch (ml_eff), not for the capacity
  // max_hap_len: half the footprint on 1 kb windows, five workgroups per CU instead of three
  u32 const ML = ml_eff;
  u32 const pw = (ML + 31) / 32 + 2;  // words per haplotype bit plane (two zero words of padding)
  u32* code = lds_vote;                                    // [ML]
  u16* head = reinterpret_cast<u16*>(code + ML);           // [kIdxCap]
  u16* next = head + kIdxCap;                              // [ML]
  u16* hist_all = next + ((ML + 1) & ~1u);                 // [4][hist_len]
  u32* hlo = reinterpret_cast<u32*>(hist_all + 4 * hist_len);  // [3][pw]
  u32* hhi = hlo + pw;
  u32* hbad = hhi + pw;
  u32* rplanes_all = hbad + pw;                            // [4 waves][3][rwords]
  u32* dpbuf_all = rplanes_all + 12 * rwords;              // [4 waves][129]
  u16* dup_pre = reinterpret_cast<u16*>(dpbuf_all + 4 * 129);  // [ML + 2]
  u16* l_rlen = dup_pre + ((ML + 4) & ~1u);  // [nr] read lengths (<= kMaxGenoRead)
  // [4 waves][left_cap] the reads a wave's eight-reads-per-trip pass leaves to the wave-wide route (bit 15: hint shortcut tried)
  u32 const left_cap = (nr + 3u) / 4u + 10u;
  u16* l_left_all = l_rlen + ((nr + 3u) & ~1u);
  size_t const hi = static_cast<size_t>(w) * A.prm.max_haps + slot;
  u32 const n = A.a.hap_len[hi];
  const u8* hb = A.a.hap_bases + hi * A.prm.max_hap_len;
  for (u32 x = threadIdx.x; x < kIdxCap / 2; x += 256) reinterpret_cast<u32*>(head)[x] = 0xFFFFFFFFu;
  for (u32 x = threadIdx.x; x < 4 * hist_len / 2; x += 256) reinterpret_cast<u32*>(hist_all)[x] = 0;
  for (u32 x = threadIdx.x; x < nr; x += 256) l_rlen[x] = static_cast<u16>(A.b.read_off[r0 + x + 1] - A.b.read_off[r0 + x]);
  if (threadIdx.x < 4) dpbuf_all[threadIdx.x * 129 + 64] = 0;
  // haplotype bases -> three bit planes (one coalesced byte load per base, wave ballots)
  for (u32 j0 = 0; j0 < pw * 32; j0 += 256) {
    u32 const j = j0 + threadIdx.x;
    u32 const e = j < n ? enc_base(hb[j]) : 0u;
    unsigned long long const blo = __ballot(e & 1u), bhi = __ballot(e & 2u), bbad = __ballot(e > 3u);
    if ((threadIdx.x & 63) == 0 && (j >> 5) + 1 < pw + 1) {
      u32 const wd = j >> 5;
      if (wd < pw) {
        hlo[wd] = static_cast<u32>(blo);
        hhi[wd] = static_cast<u32>(bhi);
        hbad[wd] = static_cast<u32>(bbad);
      }
      if (wd + 1 < pw) {
        hlo[wd + 1] = static_cast<u32>(blo >> 32);
        hhi[wd + 1] = static_cast<u32>(bhi >> 32);
        hbad[wd + 1] = static_cast<u32>(bbad >> 32);
      }
    }
  }
  __syncthreads();
  for (u32 j = threadIdx.x; j + SK <= n; j += 256) {
    bool const ok = plane11(hbad, j) == 0;
    u32 const cd = plane11(hlo, j) | (plane11(hhi, j) << 11);  // any injective code of the 11-mer will do
    code[j] = ok ? cd : 0xFFFFFFFFu;
    if (!ok) continue;
    u32 const bkt = (cd * 2654435761u) >> (32 - 12);  // kIdxCap == 4096
    // push-front into the bucket chain (order irrelevant: votes commute).  16-bit CAS via 32-bit word.
    u32* word = reinterpret_cast<u32*>(head) + (bkt >> 1);
    u32 const shift = (bkt & 1u) * 16u;
    u32 old = *word;
    while (true) {
      next[j] = static_cast<u16>((old >> shift) & 0xFFFFu);
      u32 const nw = (old & ~(0xFFFFu << shift)) | (j << shift);
      u32 const seen = atomicCAS(word, old, nw);
      if (seen == old) break;
      old = seen;
    }
  }
  __syncthreads();
  // flag the 11-mers that occur more than once in the haplotype (a vote on them is never unanimous)
  for (u32 j = threadIdx.x; j + SK <= n; j += 256) {
    u32 const cd = code[j];
    if (cd == 0xFFFFFFFFu) continue;
    u32 cnt = 0;
    for (u32 x = head[(cd * 2654435761u) >> (32 - 12)]; x != 0xFFFFu && cnt < 2; x = next[x]) cnt += same_code(code[x], cd);
    if (cnt >= 2) code[j] = cd | kCodeDup;  // (other threads compare through same_code: the flag never disturbs them)
  }
  __syncthreads();
  int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // dup_pre[j] = sum over the haplotype positions j' < j of (occurrences of the 11-mer at j' in the haplotype - 1): the votes
  // that matching those positions exactly casts on OTHER diagonals (for the hint shortcuts' vote bounds; a position whose
  // 11-mer is unique adds nothing).  sh_mmax = the most occurrences of any 11-mer: what a read 11-mer that is NOT the
  // haplotype's own at its place can vote at most.
  __shared__ u32 sh_dup[4];
  __shared__ u32 sh_mmax;
  __shared__ i32 sh_cand[12];
  if (threadIdx.x == 0) sh_mmax = 1;
  __syncthreads();
  u32 running = 0;  // (the same in every thread)
  {
    for (u32 j0 = 0; j0 <= n; j0 += 256) {
      u32 const j = j0 + threadIdx.x;
      u32 extra = 0;
      if (j + SK <= n && code[j] != 0xFFFFFFFFu && (code[j] & kCodeDup)) {
        u32 const cd = code[j];
        u32 cnt = 0;
        for (u32 x = head[((cd & ~kCodeDup) * 2654435761u) >> (32 - 12)]; x != 0xFFFFu && cnt < 255u; x = next[x]) cnt += same_code(code[x], cd);
        extra = cnt > 0 ? cnt - 1u : 0u;
        atomicMax(&sh_mmax, cnt);
      }
      u32 inc = extra;
      for (int d = 1; d < 64; d <<= 1) {
        u32 const y = __shfl_up(inc, d);
        if (lane >= d) inc += y;
      }
      if (lane == 63) sh_dup[wave] = inc;
      __syncthreads();
      u32 before = running;
      for (int x = 0; x < wave; ++x) before += sh_dup[x];
      if (j <= n) dup_pre[j] = static_cast<u16>(min(before + inc - extra, 65535u));
      running += sh_dup[0] + sh_dup[1] + sh_dup[2] + sh_dup[3];
      __syncthreads();
    }
  }
  // Hint shortcut candidates: a read mapped at window offset `hint` lies on diagonal hint - anchor of the REF haplotype
  // and, on an ALT haplotype, on that diagonal shifted by the net length change of the variants before it.
  if (threadIdx.x == 0) {
    int nc = 0;
    sh_cand[0] = 0;
    if (A.b.read_hint && !(A.prm.aln_tier & 2) && running < 60000u) {  // (a 16-bit prefix sum that saturated bounds nothing)
      ma_params_t const& P = A.prm;
      for (u32 c = 0; c < A.a.win_ncomp[w]; ++c) {
        size_t const ci = static_cast<size_t>(w) * P.max_comps + c;
        u32 const h0 = A.a.comp_hap0[ci], nh = A.a.comp_nhaps[ci];
        if (static_cast<u32>(slot) < h0 || static_cast<u32>(slot) >= h0 + nh) continue;
        u32 const h = static_cast<u32>(slot) - h0;
        sh_cand[1] = static_cast<i32>(A.a.comp_anchor[ci]);
        sh_cand[2 + nc++] = 0;
        for (u32 v = 0; v < A.v.win_nvars[w] && h > 0; ++v) {
          size_t const vi = static_cast<size_t>(w) * P.max_vars + v;
          if (A.v.var_comp[vi] != c) continue;
          u32 const al = A.v.var_hap_allele[vi * P.max_haps + h];
          if (al == 0) continue;
          i32 const sh = static_cast<i32>(A.v.var_hap_start[vi * P.max_haps + h] + A.v.alt_len[vi * P.max_alts + (al - 1)]) -
                         static_cast<i32>(A.v.var_ref_start[vi] + A.v.var_ref_len[vi]);
          bool seen = false;
          for (int x = 0; x < nc; ++x) seen = seen || sh_cand[2 + x] == sh;
          if (seen) continue;
          if (nc >= 8) {  // too many distinct shifts: not worth trying
            nc = 0;
            break;
          }
          sh_cand[2 + nc++] = sh;
        }
        break;
      }
    }
    sh_cand[0] = nc;
  }
  __syncthreads();
#if defined(MA_VOTE_STOP) && MA_VOTE_STOP == 1  // (developer timing build, tools/dbg/vote_phases.sh: the set-up alone; results invalid)
  return;
#endif
#ifdef MA_PROFILE
  __shared__ unsigned long long sh_prof[4][8];
  if (lane < 8) sh_prof[wave][lane] = 0;
  unsigned long long const k_t0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) atomicAdd(&g_vprof[15], k_t0 - k_tstart);  // workgroup set-up: planes, seed index, repeat flags, prefix counts
  u32 hap_amb = 0;
  for (u32 x = lane; x < pw; x += 64) hap_amb |= hbad[x];
  hap_amb = __ballot(hap_amb != 0) ? 1u : 0u;
  HapIdx const ix{head, next, code, hlo, hhi, hbad, rplanes_all + static_cast<size_t>(wave) * 3 * rwords, rwords,
                  dpbuf_all + wave * 129, hap_amb, dup_pre, sh_cand, sh_prof[wave]};
#else
  u32 hap_amb = 0;
  for (u32 x = lane; x < pw; x += 64) hap_amb |= hbad[x];
  hap_amb = __ballot(hap_amb != 0) ? 1u : 0u;
  HapIdx const ix{head, next, code, hlo, hhi, hbad, rplanes_all + static_cast<size_t>(wave) * 3 * rwords, rwords,
                  dpbuf_all + wave * 129, hap_amb, dup_pre, sh_cand};
#endif
#ifdef MA_PROFILE_TRIPS  // (developer build: where a trip of eight reads spends its time -- [0] candidates [1] group vote [2] wave-wide route)
  unsigned long long const k_t0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) atomicAdd(&g_vprof[15], k_t0 - k_tstart);
  unsigned long long tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0, tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0;
#define TRIP_STAMP(v) unsigned long long const v = __builtin_amdgcn_s_memtime()
#else
#define TRIP_STAMP(v) do {} while (0)
#endif
  u16* hist = hist_all + static_cast<size_t>(wave) * hist_len;
  u32 const npw = 3u * rwords;
  bool const hinted = A.b.read_hint != nullptr;
  u16* l_left = l_left_all + static_cast<size_t>(wave) * left_cap;
  u32 nleft = 0;  // (wave-uniform)
  bool const grouped = rwords <= 8 && ix.cand[0] > 0 && !ix.hap_amb && hinted && nr > 0 && nr < 32768u;
  if (grouped) {
    // EIGHT reads per trip: the hint shortcut needs a lane per 32 bases -- five lanes of the wave for a 150-base read -- and
    // two reads in three end there.  Lane 8 g + x holds word x of read g's three planes (reads of up to 256 bases); the
    // candidate diagonals of all eight reads are tried side by side with 8-lane reductions, every group's first lane
    // writes its own record.  Only the reads that are left go through the wave-wide vote, one after the other, their
    // plane words brought into the lane-l-holds-word-l layout by shuffles.  (One read per trip: the shortcut, a third of
    // this kernel's time, ran on 5 of 64 lanes.)
    u32 const g = static_cast<u32>(lane) >> 3, x = static_cast<u32>(lane) & 7u;
    auto group_read = [&](u32 q0) -> u32 { return static_cast<u32>(wave) + 4u * (q0 + g); };
    auto fetch_words = [&](u32 ri, u32* wl, u32* wh, u32* wb, i32* hn) {
      bool const ok = ri < nr && x < rwords;
      const u32* pl = A.ws.read_planes + static_cast<size_t>(r0 + min(ri, nr - 1)) * plane_stride(rwords);
      u32 const a0 = pl[min(x, rwords - 1)], a1 = pl[rwords + min(x, rwords - 1)], a2 = pl[2 * rwords + min(x, rwords - 1)];
      *wl = ok ? a0 : 0u;
      *wh = ok ? a1 : 0u;
      *wb = ok ? a2 : 0u;
      *hn = ri < nr ? A.b.read_hint[r0 + ri] : MA_NO_HINT;
    };
    u32 nwl, nwh, nwb;
    i32 nhn;
    fetch_words(group_read(0), &nwl, &nwh, &nwb, &nhn);
    for (u32 q0 = 0; static_cast<u32>(wave) + 4u * q0 < nr; q0 += 8) {
      u32 const ri = group_read(q0);
      u32 const wl = nwl, wh = nwh, wb = nwb;
      i32 const hint = nhn;
      fetch_words(group_read(q0 + 8), &nwl, &nwh, &nwb, &nhn);  // (the next eight reads' words: in flight under this trip)
      u64 const p = p0 + ri;
      bool const valid = ri < nr && p >= A.pair0 && p < A.pair0 + A.npairs;
      i32 const m = valid ? static_cast<i32>(l_rlen[ri]) : 0;
      bool can = valid && m >= SK && m <= 256 && hint != MA_NO_HINT;  // (longer reads: the wave-wide route tries the shortcut itself)
      bool const tried = can;
      bool settled = false;
      bool read_n;
      {  // an N in the read: general route (the per-read code leaves its candidate loop at the first in-range candidate)
        u32 const b_ = grp8_or(wb);
        read_n = b_ != 0;
        if (read_n) can = false;
      }
      i32 const ncand = ix.cand[0];
      i32 const mmax = static_cast<i32>(sh_mmax);
      TRIP_STAMP(ts0);
      for (int cx = 0; cx < ncand; ++cx) {
        if (__ballot(can && !settled) == 0ull) break;
        i32 const c = hint - ix.cand[1] + ix.cand[2 + cx];
        // the read inside the haplotype on this diagonal -- or overhanging ONE of its ends by o bases (rows [qs, qe) overlap)
        i32 const o_left = c < 0 ? -c : 0, o_right = c + m > n ? c + m - n : 0;
        i32 const qs = o_left, qe = m - o_right;
        bool const inr = can && !settled && o_left == 0 && o_right == 0;
        bool const ovr = can && !settled && ((o_left == 0) != (o_right == 0)) && qe - qs >= SK && A.prm.max_cigar >= 2;
        u32 mism = 0;
        i32 const i = 32 * static_cast<i32>(x);
        if ((inr || ovr) && i < m) {
          i32 const lo = max(qs - i, 0), hi = min(qe - i, 32);  // overlap bits of this word: [lo, hi)
          if (hi > lo) {
            u32 const vmask = (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
            i32 const hp = c + i + lo;  // >= 0
            u32 const hl = plane32(ix.hlo, hp) << lo, hh = plane32(ix.hhi, hp) << lo;
            mism = __popc(((wl ^ hl) | (wh ^ hh)) & vmask);
          }
        }
        i32 const X = static_cast<i32>(grp8_sum(mism));
        if (inr && X <= 2) {
          i32 const D = static_cast<i32>(ix.dup_pre[c + m - SK + 1]) - static_cast<i32>(ix.dup_pre[c]);
          i32 const S0 = m - 5 * X;
          if (D + 22 * X + 10 < m && S0 >= A.prm.min_aln_score && D < 60000) {
            settled = true;
            if (x == 0) {
              rec_store(rec_at(A.ws, p), S0, c, c + m, 0, m, 1u, static_cast<u32>(m) << 4, 0u, 0u, 0u);
              A.ws.centre[p - A.pair0] = 0x7FFFFFFE;
            }
          }
        } else if (ovr && X <= 2) {
          // Certificate (II) without the seed index (round 5; timing builds put 83 % of this kernel into the wave-wide route and
          // a quarter of the pairs there only because the read hangs over a haplotype end).  With L0 = qe - qs overlap rows,
          // X <= 2 mismatches and nothing ambiguous, diagonal c holds >= L0 - 10 - 11 X votes.  Every vote NOT on c comes from
          // a read 11-mer that (a) lies in the overlap and equals the haplotype's own 11-mer at its place -- it votes elsewhere
          // once per further occurrence of that 11-mer: D in total, from the prefix sums of (occurrences - 1) -- or (b) overlaps a
          // mismatch (<= 11 X positions) or the overhang (o positions): at most mmax votes each, mmax = the most occurrences of
          // any 11-mer of the haplotype.  So V_off <= U = D + (11 X + o + 10) mmax.  If L0 - 10 - 11 X > U, c is the strictly
          // most-voted diagonal -- the one the wave-wide route would examine -- and an anchor; if moreover 11 S0 > 6 m + 50 + 5 U,
          // its certificate (II) holds with the true V_off <= U: the same record, without the votes.  Anything else takes the
          // wave-wide route as before.
          i32 const L0 = qe - qs, o = o_left + o_right;
          i32 const D = static_cast<i32>(ix.dup_pre[c + qe - SK + 1]) - static_cast<i32>(ix.dup_pre[c + qs]);
          i32 const U = D + (11 * X + o + 10) * mmax;
          i32 const S0 = L0 - 5 * X;
          if (S0 >= A.prm.min_aln_score && L0 - 10 - 11 * X > U && 11 * S0 > 6 * m + 50 + 5 * U && D < 60000 && mmax < 255) {
            settled = true;
            if (x == 0) {
              u32 const ops_m = static_cast<u32>(L0) << 4;
              if (qs > 0) rec_store(rec_at(A.ws, p), S0, c + qs, c + qe, qs, qe, 2u, (static_cast<u32>(qs) << 4) | 4u, ops_m, 0u, 0u);
              else rec_store(rec_at(A.ws, p), S0, c + qs, c + qe, qs, qe, 2u, ops_m, (static_cast<u32>(m - qe) << 4) | 4u, 0u, 0u);
              A.ws.centre[p - A.pair0] = 0x7FFFFFFE;
            }
          }
        }
      }
      TRIP_STAMP(ts1);
      TRIP_STAMP(ts2);
      // the reads that are left go on this wave's list: the wave-wide route runs in a loop of its own below (inlined into this
      // loop it shared the trip's registers: the prefetched words were reloaded from scratch behind a vmcnt(0) every trip)
#if defined(MA_VOTE_STOP) && MA_VOTE_STOP == 2  // (developer timing build: the shortcut trips without the wave-wide route; results invalid)
      unsigned long long const todo = 0ull;
#else
      unsigned long long const todo = __ballot(valid && !settled && x == 0);
#endif
      if ((todo >> lane) & 1ull)
        l_left[nleft + static_cast<u32>(__popcll(todo & ((1ull << lane) - 1ull)))] = static_cast<u16>(ri | (tried ? 0x8000u : 0u));
      nleft += static_cast<u32>(__popcll(todo));
#ifdef MA_PROFILE_TRIPS
      tp0 += ts2 - ts0;
      tp3 += 1;
#endif
    }
    __builtin_amdgcn_wave_barrier();
  }
#ifndef MA_NO_GROUP_VOTE
  if (grouped && nleft > 0 && static_cast<i32>(n) >= SK) {
    // ---- the UNANIMOUS vote, eight listed reads side by side (round 5) ---------------------------------------------------
    // What is listed -- reads without a usable hint, with more than two mismatches, with an indel, or hanging far over a
    // haplotype end: 28 % of the pairs -- took the wave-wide route one after the other, 83 % of this kernel's time.  Nine in
    // ten of those votes are unanimous: every 11-mer that is found lies on ONE diagonal and none of them is a repeat.  That
    // case needs no histogram.  Lane 8 g + x holds word x of listed read g's planes again; the group's eight lanes look up
    // CONTIGUOUS runs of their read's positions (one 64-bit window per plane and lane, two shuffles each -- the kernel is
    // bound by the CU's LDS pipe, every one of its 20 waves gathering from the seed index: LDS instructions are what a trip
    // costs), eight positions at a time: their bucket heads, then their first entries, are independent accesses in flight,
    // and the longer bucket chains of all eight advance together.  The group keeps the count and the extreme diagonals; when
    // the vote is unanimous (or empty) its first lane settles the pair exactly as vote_pair's unanimous branch does: best =
    // the count, no runner-up, no vote off c, anchors = {c} iff the count reaches kMinChainVotes.  A repeat, a second
    // diagonal or an N keeps the read on the list for the wave-wide route.  (Run inside the trips above, on the two or
    // three unsettled reads of each trip with the other groups' lanes idle, this vote cost as much as the route it replaces.)
    u32 const g = static_cast<u32>(lane) >> 3, x = static_cast<u32>(lane) & 7u;
    int const gl = lane & ~7;
    u32 const nA = nleft;
    u32 nB = 0;
    auto fetch_item = [&](u32 q, u32* wl, u32* wh, u32* wb, u32* it) {
      u32 const idx = q + g;
      bool const ok = idx < nA;
      u32 const e = ok ? l_left[idx] : 0u;
      const u32* pl = A.ws.read_planes + static_cast<size_t>(r0 + (e & 0x7FFFu)) * plane_stride(rwords);
      u32 const xx = min(x, rwords - 1);
      u32 const a0 = pl[xx], a1 = pl[rwords + xx], a2 = pl[2 * rwords + xx];
      bool const okw = ok && x < rwords;
      *wl = okw ? a0 : 0u;
      *wh = okw ? a1 : 0u;
      *wb = okw ? a2 : 0u;
      *it = ok ? (e | 0x10000u) : 0u;
    };
    u32 nwl, nwh, nwb, nit;
    fetch_item(0, &nwl, &nwh, &nwb, &nit);
    for (u32 q = 0; q < nA; q += 8) {
      TRIP_STAMP(tg0);
      u32 const wl = nwl, wh = nwh, it = nit;
      u32 bad = nwb;
      fetch_item(q + 8, &nwl, &nwh, &nwb, &nit);  // (the next eight listed reads: in flight under this trip)
      bad = grp8_or(bad);
      bool const valid = (it & 0x10000u) != 0u;
      u32 const ri = it & 0x7FFFu;
      u64 const p = p0 + ri;
      i32 const m = valid ? static_cast<i32>(l_rlen[ri]) : 0;
      bool const gv = valid && bad == 0u && m >= SK;  // (m <= 32 (rwords - 2) <= 192 here)
      // this lane's run of positions: [s, s + pr) of the read's np = m - 10
      i32 const np = m - SK + 1;
      i32 const pr = gv ? (np + 7) / 8 : 0;
      i32 const s = static_cast<i32>(x) * pr;
      u32 const off = static_cast<u32>(s) & 31u;
      int const k = gl + (s >> 5);  // (s <= 7 * 23: the window's two words sit in lanes gl + k, gl + k + 1 <= gl + 6)
      u64 const lo64 = static_cast<u64>(static_cast<u32>(__shfl(static_cast<int>(wl), k))) | (static_cast<u64>(static_cast<u32>(__shfl(static_cast<int>(wl), k + 1))) << 32);
      u64 const hi64 = static_cast<u64>(static_cast<u32>(__shfl(static_cast<int>(wh), k))) | (static_cast<u64>(static_cast<u32>(__shfl(static_cast<int>(wh), k + 1))) << 32);
      u32 cnt = 0;
      // this lane's extreme diagonals and the votes ON each (two diagonals without a repeat are settled here too: below)
      i32 dmin = 0x7FFFFFFF, dmax = -0x7FFFFFFF;
      u32 clo = 0, chi = 0;
      auto add_votes = [&](i32 dl, u32 nl, i32 dh, u32 nh) {  // nl votes on dl <= dh, nh on dh (the same votes when dl == dh)
        if (dl < dmin) {
          dmin = dl;
          clo = nl;
        } else if (dl == dmin) {
          clo += nl;
        }
        if (dh > dmax) {
          dmax = dh;
          chi = nh;
        } else if (dh == dmax) {
          chi += nh;
        }
      };
      u32 multi = 0;
      // (the loop is bound by VALU issue -- ~60 instructions per lane and position in its first form, five waves per SIMD:
      //  every load below is unconditional with a harmless address, every update a select, the 64-bit shifts one per batch)
#if defined(MA_VOTE_STOP) && MA_VOTE_STOP == 4  // (developer timing build: the group vote without its lookups; results invalid)
      i32 const nrun = 0;
#else
      i32 const nrun = min(pr, np - s);  // this lane's positions: t < nrun (<= 0: none)
#endif
      for (i32 t0 = 0; t0 < 24; t0 += 8) {
        i32 const nact = nrun - t0;
        if (__ballot(nact > 0) == 0ull) break;
        u32 const shb = off + static_cast<u32>(t0);  // <= 31 + 16; bits shb + tt .. shb + tt + 10 <= 63 for every position of the run
        u32 const wlo = static_cast<u32>(lo64 >> shb), whi = static_cast<u32>(hi64 >> shb);
        u32 cdv[8], jv[8], ev[8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          cdv[tt] = ((wlo >> tt) & 0x7FFu) | (((whi >> tt) & 0x7FFu) << 11);
          jv[tt] = head[(cdv[tt] * 2654435761u) >> (32 - 12)];
        }
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          jv[tt] = tt < nact ? jv[tt] : 0xFFFFu;
          ev[tt] = code[jv[tt] != 0xFFFFu ? jv[tt] : 0u];
        }
        u32 pend = 0, hits = 0, cntb = 0;
        u32 bmin = 0xFFFFFFFFu, bmax = 0u;  // extremes of j + 8 - tt over this batch's first-probe hits
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          bool const have = jv[tt] != 0xFFFFu, same = same_code(ev[tt], cdv[tt]);
          bool const hit = have && same;
          pend |= (have && !same) ? (1u << tt) : 0u;
          hits |= hit ? (1u << tt) : 0u;
          u32 const vb = jv[tt] + static_cast<u32>(8 - tt);
          bmin = min(bmin, hit ? vb : 0xFFFFFFFFu);
          bmax = max(bmax, hit ? vb : 0u);
          cntb += hit ? 1u : 0u;
          multi |= hit ? ev[tt] : 0u;  // (only bit 31 is looked at)
        }
        cnt += cntb;
        u32 nlo = cntb, nhi = cntb;
        if (__ballot(bmax != 0u && bmin != bmax) != 0ull) {  // a lane whose eight positions straddle the indel: votes per extreme
          nlo = nhi = 0;
#pragma unroll
          for (int tt = 0; tt < 8; ++tt) {
            u32 const vb = jv[tt] + static_cast<u32>(8 - tt);
            nlo += ((hits >> tt) & 1u) && vb == bmin ? 1u : 0u;
            nhi += ((hits >> tt) & 1u) && vb == bmax ? 1u : 0u;
          }
        }
        if (bmax != 0u) {
          i32 const base = s + t0 + 8;
          add_votes(static_cast<i32>(bmin) - base, nlo, static_cast<i32>(bmax) - base, nhi);
        }
        // Buckets shared with other 11-mers (one probe in ten): every lane walks ITS pending chains one after the other --
        // a handful of LDS instructions for the wave.  (All eight positions in lock step issued sixteen, mostly empty, per
        // chain step.)
#if defined(MA_VOTE_STOP) && MA_VOTE_STOP == 5  // (developer timing build: first probes only; results invalid)
        pend = 0;
#endif
        while (pend != 0) {
          u32 const tt = static_cast<u32>(__builtin_ctz(pend));
          pend &= pend - 1u;
          u32 j = jv[0], cd = cdv[0];
#pragma unroll
          for (u32 k = 1; k < 8; ++k) {
            j = tt == k ? jv[k] : j;
            cd = tt == k ? cdv[k] : cd;
          }
          u32 e = 0;
          do {
            j = next[j];
            if (j == 0xFFFFu) break;
            e = code[j];
          } while (!same_code(e, cd));
          if (j != 0xFFFFu) {
            i32 const d = static_cast<i32>(j) - (s + t0 + static_cast<i32>(tt));
            ++cnt;
            add_votes(d, 1u, d, 1u);
            multi |= e;
          }
        }
      }
      multi &= kCodeDup;
      TRIP_STAMP(tgA);
      cnt = grp8_sum(cnt);
#if defined(MA_VOTE_STOP) && MA_VOTE_STOP == 6  // (developer timing build: the lookups run, every listed read is then called 'no shared 11-mer'; results invalid)
      if (cnt != 0x12345678u) cnt = 0;
#endif
      i32 const lmin = dmin, lmax = dmax;
      dmin = grp8_min(dmin);
      dmax = grp8_max(dmax);
      multi = grp8_or(multi);
      bool const unan = gv && multi == 0 && (cnt == 0 || dmin == dmax);
      // TWO diagonals, no repeat (a read across an indel of the other allele, both sides long enough to vote): what the
      // histogram route would find is two counters -- na votes on da = dmin, nb on db = dmax, nothing else (every vote
      // is on one of the group's two extremes iff the votes on them add up to all of them).  Most-voted diagonal, ties to
      // the smaller; runner-up = the other; anchors: a diagonal with >= kMinChainVotes votes within reach K of it.
      u32 const na = grp8_sum(lmin == dmin ? clo : (lmax == dmin ? chi : 0u));
      u32 const nb = grp8_sum(lmax == dmax ? chi : (lmin == dmax ? clo : 0u));
      bool const two = gv && multi == 0 && cnt != 0 && dmin != dmax && na + nb == cnt;
      i32 const Kq = m - A.prm.min_aln_score - GO > 0 ? (m - A.prm.min_aln_score - GO) / GE : 0;
      bool const near2 = two && dmax - dmin <= Kq;
      bool const ancA = two && na + (near2 ? nb : 0u) >= kMinChainVotes, ancB = two && nb + (near2 ? na : 0u) >= kMinChainVotes;
      bool const anchored = (unan && cnt >= kMinChainVotes) || ancA || ancB;
      // mismatches of the gapless path on the most-voted diagonal c (vote_pair's rule: only when the read is inside or
      // overhangs ONE end)
      i32 const c2 = two ? (na >= nb ? dmax : dmin) : 0;  // the runner-up
      i32 const c = anchored ? (two ? (na >= nb ? dmin : dmax) : dmin) : 0;
      i32 const o_left = c < 0 ? -c : 0, o_right = c + m > static_cast<i32>(n) ? c + m - static_cast<i32>(n) : 0;
      i32 const qs = o_left, qe = m - o_right;
      u32 mism = 0;
      {
        i32 const i = 32 * static_cast<i32>(x);
        if (anchored && (o_left == 0 || o_right == 0) && i < m) {
          i32 const lo = max(qs - i, 0), hi = min(qe - i, 32);  // overlap bits of this word: [lo, hi)
          if (hi > lo) {
            u32 const vmask = (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
            i32 const hp = c + i + lo;  // >= 0
            u32 const hl = plane32(ix.hlo, hp) << lo, hh = plane32(ix.hhi, hp) << lo;
            mism = __popc(((wl ^ hl) | (wh ^ hh)) & vmask);
          }
        }
      }
      mism = grp8_sum(mism);
      // the one-gap lower bound between the two diagonals (see vote_pair), on the group's eight words
      i32 lb_two = -(1 << 20);
      i32 const vmin2 = ancA ? dmin : dmax, vmax2 = ancB ? dmax : dmin;  // extreme anchors of a two-diagonal vote
      if (__ballot(two && anchored) != 0ull) {
        bool const okc = two && anchored && c >= 0 && c + m <= static_cast<i32>(n) && c2 >= 0 && c2 + m <= static_cast<i32>(n) &&
                         c >= vmin2 && c <= vmax2 && c2 >= vmin2 && c2 <= vmax2;
        i32 const i = 32 * static_cast<i32>(x);
        u32 xa = 0, xb = 0;
        if (okc && i < m) {
          u32 const vmask = m - i >= 32 ? 0xFFFFFFFFu : ((1u << (m - i)) - 1u);
          xa = ((wl ^ plane32(ix.hlo, c + i)) | (wh ^ plane32(ix.hhi, c + i))) & vmask;
          xb = ((wl ^ plane32(ix.hlo, c2 + i)) | (wh ^ plane32(ix.hhi, c2 + i))) & vmask;
        }
        auto one_gap = [&](u32 first, u32 second, i32 a, i32 b) -> i32 {
          i32 const t = min(m, grp8_min(first ? i + static_cast<i32>(__builtin_ctz(first)) : 0x7FFFFFFF));
          i32 const sgap = b > a ? b - a : a - b;
          i32 const start = b > a ? t : t + sgap;  // the first row on the second diagonal
          i32 const rel = start - i;
          u32 const from = rel <= 0 ? 0xFFFFFFFFu : (rel >= 32 ? 0u : ~((1u << rel) - 1u));
          i32 const xs = static_cast<i32>(grp8_sum(static_cast<u32>(__popc(second & from))));
          return start > m ? -(1 << 20) : (b > a ? m : m - sgap) - 5 * xs - (GO + GE * sgap);
        };
        i32 const g1 = one_gap(xa, xb, c, c2), g2 = one_gap(xb, xa, c2, c);
        if (okc) lb_two = max(g1, g2);
      }
      TRIP_STAMP(tgB);
      u32 act = kVoteNoHit;
      if ((unan || two) && x == 0) {
        u64 const lp = p - A.pair0;
        if (!anchored) {  // no shared 11-mer, or fewer than a chain needs: no hit
          A.ws.centre[lp] = 0x7FFFFFFF;
          write_no_hit(A, lp);
        } else if (unan) {
          act = vote_settle(A, lp, PairId{w, r0 + ri, static_cast<u32>(slot)}, m, static_cast<i32>(n), c, c, c, Kq, static_cast<i32>(mism), false,
                            false, false, 0, 0, 0u, 0u, 0u);
        } else {
          u32 const v2 = min(na, nb);
          u32 const far = static_cast<u32>(dmax - dmin);
          act = vote_settle(A, lp, PairId{w, r0 + ri, static_cast<u32>(slot)}, m, static_cast<i32>(n), c, vmin2, vmax2, Kq, static_cast<i32>(mism),
                            false, false, false, static_cast<i32>(v2), static_cast<i32>(v2), far > 8u ? v2 : 0u, far > 16u ? v2 : 0u,
                            far > 24u ? v2 : 0u, lb_two);
        }
      }
#if defined(MA_VOTE_STOP) && MA_VOTE_STOP == 7  // (developer timing build: the group vote's DP pairs are not queued; results invalid)
      act = kVoteFast;
#endif
      TRIP_STAMP(tgC);
      // the DP pairs of this trip (at most eight) into the wave's buffer: it holds < 64 on entry
      unsigned long long const dpm = __ballot(act < kVoteFast);
      if (dpm != 0ull) {
        u32 const ndp = static_cast<u32>(__popcll(dpm));
        if (ix.dpbuf[64] + ndp > 64u) vote_flush_dp(A, ix, lane);
        u32 const base = ix.dpbuf[64];
        __builtin_amdgcn_wave_barrier();
        if (act < kVoteFast) {
          u32 const at = base + static_cast<u32>(__popcll(dpm & ((1ull << lane) - 1ull)));
          ix.dpbuf[at] = static_cast<u32>(p - A.pair0);
          ix.dpbuf[65 + at] = act;
        }
        if (lane == 0) ix.dpbuf[64] = base + ndp;
        __builtin_amdgcn_wave_barrier();
        if (ix.dpbuf[64] == 64u) vote_flush_dp(A, ix, lane);
      }
      // what stays listed (compacted in place: nB <= q, and the entries up to q + 15 have been read)
      unsigned long long const stay = __ballot(valid && !(unan || two) && x == 0);
#ifdef MA_PROFILE_TRIPS  // (census of what stays listed: a repeat / two or more diagonals without a repeat / not voted here at all)
      if (lane == 0) {
        atomicAdd(&g_vprof[9], static_cast<unsigned long long>(__popcll(__ballot(valid && gv && multi != 0 && x == 0))));
        atomicAdd(&g_vprof[10], static_cast<unsigned long long>(__popcll(__ballot(valid && gv && multi == 0 && !unan && !two && x == 0))));
        atomicAdd(&g_vprof[11], static_cast<unsigned long long>(__popcll(__ballot(valid && !gv && x == 0))));
      }
#endif
      if ((stay >> lane) & 1ull) l_left[nB + static_cast<u32>(__popcll(stay & ((1ull << lane) - 1ull)))] = static_cast<u16>(it & 0xFFFFu);
      nB += static_cast<u32>(__popcll(stay));
#ifdef MA_PROFILE_TRIPS
      TRIP_STAMP(tgD);
      tp1 += tgD - tg0;
      tq0 += tgA - tg0;
      tq1 += tgB - tgA;
      tq2 += tgC - tgB;
      tq3 += tgD - tgC;
      tp4 += 1;
#endif
    }
    __builtin_amdgcn_wave_barrier();
    nleft = nB;
  }
#endif
#if defined(MA_VOTE_STOP) && MA_VOTE_STOP >= 3  // (developer timing builds: no wave-wide route after the group vote; results invalid)
  nleft = 0;
#endif
  {
    // ---- the wave-wide route: the listed reads (or, without the trips, all of this wave's reads), one after the other ----
    // software pipeline: the next reads' plane words (lane l < 3 rwords holds word l) and mapping hints are in flight while
    // the current read is voted -- kPF reads ahead
    TRIP_STAMP(tw0);
    u32 const nloop = grouped ? nleft : (nr > static_cast<u32>(wave) ? (nr - static_cast<u32>(wave) + 3u) / 4u : 0u);
    auto item = [&](u32 idx) -> u32 {  // read index, bit 31: the hint shortcut was tried
      if (!grouped) return static_cast<u32>(wave) + 4u * idx;
      u32 const e = l_left[idx];
      return (e & 0x7FFFu) | ((e >> 15) << 31);
    };
    auto fetch = [&](u32 ri) -> u32 {
      return static_cast<u32>(lane) < npw ? A.ws.read_planes[static_cast<size_t>(r0 + ri) * plane_stride(rwords) + lane] : 0u;
    };
    constexpr u32 kPF = MA_VOTE_PF;
    u32 pf[kPF], it[kPF];
    i32 hf[kPF];
#pragma unroll
    for (u32 x = 0; x < kPF; ++x) {
      it[x] = x < nloop ? item(x) : 0u;
      u32 const ri = it[x] & 0x7FFFFFFFu;
      pf[x] = x < nloop ? fetch(ri) : 0u;
      hf[x] = (hinted && x < nloop) ? A.b.read_hint[r0 + ri] : MA_NO_HINT;
    }
    for (u32 idx = 0; idx < nloop; ++idx) {
      u32 const ia = idx + kPF;
      u32 const itn = ia < nloop ? item(ia) : 0u;
      u32 const pn = ia < nloop ? fetch(itn & 0x7FFFFFFFu) : 0u;
      i32 const hn = (hinted && ia < nloop) ? A.b.read_hint[r0 + (itn & 0x7FFFFFFFu)] : MA_NO_HINT;
      u32 const ri = it[0] & 0x7FFFFFFFu;
      u64 const p = p0 + ri;
      if (p >= A.pair0 && p < A.pair0 + A.npairs)
        vote_pair(A, p - A.pair0, PairId{w, r0 + ri, static_cast<u32>(slot)}, ix, hist, lane,
                  static_cast<i32>(l_rlen[ri]), pf[0], hf[0], static_cast<i32>(n), (it[0] >> 31) != 0u);
#pragma unroll
      for (u32 x = 0; x + 1 < kPF; ++x) {
        pf[x] = pf[x + 1];
        hf[x] = hf[x + 1];
        it[x] = it[x + 1];
      }
      pf[kPF - 1] = pn;
      hf[kPF - 1] = hn;
      it[kPF - 1] = itn;
    }
#ifdef MA_PROFILE_TRIPS
    TRIP_STAMP(tw1);
    tp2 += tw1 - tw0;
#endif
  }
  // What the four waves still buffer goes out together: ONE returning atomic per workgroup and one per key present.  (A
  // flush per wave put ~1.5 M atomics per step on the one cache line of dp_count -- each wave's handful of pairs, its
  // returning add and three or four key counts -- and the kernel, timing builds say, spent 2 of its 5.6 ms queueing for
  // that line.)
  {
    __shared__ u32 sh_kh[36];
    __shared__ u32 sh_dpbase;
    if (threadIdx.x < 36) sh_kh[threadIdx.x] = 0;
    __syncthreads();
    u32 const c0 = dpbuf_all[64], c1 = dpbuf_all[129 + 64], c2 = dpbuf_all[2 * 129 + 64], c3 = dpbuf_all[3 * 129 + 64];
    u32 const total = c0 + c1 + c2 + c3;
    if (total != 0) {  // (workgroup-uniform)
      u32 const mine = wave == 0 ? c0 : (wave == 1 ? c1 : (wave == 2 ? c2 : c3));
      u32 const before = wave == 0 ? 0u : (wave == 1 ? c0 : (wave == 2 ? c0 + c1 : c0 + c1 + c2));
      bool const have = static_cast<u32>(lane) < mine;
      u32 const entry = have ? dpbuf_all[wave * 129 + lane] : 0u;
      if (have) atomicAdd(&sh_kh[dpbuf_all[wave * 129 + 65 + lane]], 1u);
      if (threadIdx.x == 0) sh_dpbase = atomicAdd(A.ws.dp_count, total);
      __syncthreads();
      if (have) A.ws.dp_list[sh_dpbase + before + static_cast<u32>(lane)] = entry;
      if (threadIdx.x < 36 && sh_kh[threadIdx.x] != 0) atomicAdd(&A.ws.dp_count[4 + threadIdx.x], sh_kh[threadIdx.x]);
    }
  }
#ifdef MA_PROFILE_TRIPS
  if (lane == 0) {
    atomicAdd(&g_vprof[0], tp0);
    atomicAdd(&g_vprof[1], tp1);
    atomicAdd(&g_vprof[2], tp2);
    atomicAdd(&g_vprof[7], tp3);
    atomicAdd(&g_vprof[8], tp4);
    atomicAdd(&g_vprof[3], tq0);
    atomicAdd(&g_vprof[4], tq1);
    atomicAdd(&g_vprof[5], tq2);
    atomicAdd(&g_vprof[6], tq3);
    atomicAdd(&g_vprof[14], __builtin_amdgcn_s_memtime() - k_t0);
  }
#endif
#ifdef MA_PROFILE
  __builtin_amdgcn_wave_barrier();
  if (lane < 8) atomicAdd(&g_vprof[lane], sh_prof[wave][lane]);
  if (lane == 0) atomicAdd(&g_vprof[14], __builtin_amdgcn_s_memtime() - k_t0);
#endif
}

__device__ __forceinline__ void vote_pair(GArgs const& A, u64 lp, PairId id, HapIdx ix, u16* hist, int lane, i32 m, u32 pre, i32 hint, i32 n,
                                          bool shortcut_tried) {
  i32 const nd = m + n + 1;  // (n = the haplotype's length, held by the caller)  // diagonals d in [-m, n] -> hist[d + m]
  i32 const Kr = m - A.prm.min_aln_score - GO > 0 ? (m - A.prm.min_aln_score - GO) / GE : 0;  // reach K of the search region
  const u16* head = ix.head;
  const u16* next = ix.next;
  const u32* code = ix.code;
  VPROF_T0();
  // the read's three bit planes (k_read_planes) into this wave's LDS slot: [3][rwords], one word per lane
  u32* rlo = ix.rplanes;
  u32* rhi = rlo + ix.rwords;
  u32* rbad = rhi + ix.rwords;
  if (static_cast<u32>(lane) < 3u * ix.rwords) rlo[lane] = pre;
  __builtin_amdgcn_wave_barrier();
  // ---- hint shortcut: certificate (I) without the seed index --------------------------------------------------------
  // Try the diagonals the read's mapped position suggests.  On such a diagonal c (read inside the haplotype, X <= 2
  // mismatches, nothing ambiguous) every vote for ANOTHER diagonal comes from a read position whose 11-mer either
  // overlaps a mismatch (<= 11 X positions) or equals a haplotype 11-mer that occurs more than once (D of the m - 10
  // 11-mers under the read, from the prefix counts of the repeat flags): no diagonal but c holds more than D + 11 X
  // votes, which is all certificate (I) asks of the vote histogram -- D + 11 X + 10 + 11 X < m makes P0 the unique
  // optimum.  c itself holds >= m - 10 - 11 X >= 4 votes, so it is an anchor and P0 lies in the search region.  A wrong
  // or missing hint only means that no candidate passes and the pair takes the general route: results never depend on it.
  if (!shortcut_tried && ix.cand[0] > 0 && !ix.hap_amb && m >= SK && m <= 2048) {
    if (hint != MA_NO_HINT) {  // (A.b.read_hint[id.r], fetched by the caller a read ahead)
      for (int x = 0; x < ix.cand[0]; ++x) {
        i32 const c = hint - ix.cand[1] + ix.cand[2 + x];
        if (c < 0 || c + m > n) continue;
        u32 mism = 0, bad = 0;
        {
          i32 const i = 32 * lane;
          if (i < m) {
            i32 const hi = min(m - i, 32);
            u32 const valid = hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u);
            u32 const hl = plane32(ix.hlo, c + i), hh = plane32(ix.hhi, c + i);
            bad = rbad[i >> 5];
            mism = __popc(((rlo[i >> 5] ^ hl) | (rhi[i >> 5] ^ hh)) & valid);
          }
        }
        if (__ballot(bad != 0)) break;  // an N in the read: general route
        for (int off = m <= 256 ? 4 : 32; off > 0; off >>= 1) mism += __shfl_xor(mism, off);  // words live in lanes < m / 32
        i32 const X = static_cast<i32>(__builtin_amdgcn_readfirstlane(mism));
        if (X > 2) continue;
        i32 const D = static_cast<i32>(ix.dup_pre[c + m - SK + 1]) - static_cast<i32>(ix.dup_pre[c]);
        i32 const S0 = m - 5 * X;
        if (D + 22 * X + 10 < m && S0 >= A.prm.min_aln_score && D < 60000) {
          if (lane == 0) {
            rec_store(rec_at(A.ws, A.pair0 + lp), S0, c, c + m, 0, m, 1u, static_cast<u32>(m) << 4, 0u, 0u, 0u);
            A.ws.centre[lp] = 0x7FFFFFFE;
          }
          return;
        }
      }
    }
  }
  // this lane's read positions i = lane, lane + 64, ... and their 11-mer codes (0xFFFFFFFF: none)
  constexpr int kPos = 4;  // the first 256 read positions keep their codes in registers
  u32 cds[kPos];
  bool const seeded = m >= SK && n >= SK;
  auto code_at = [&](i32 i) -> u32 {
    if (!seeded || i + SK > m) return 0xFFFFFFFFu;
    if (plane11(rbad, i) != 0) return 0xFFFFFFFFu;
    return plane11(rlo, i) | (plane11(rhi, i) << 11);
  };
#pragma unroll
  for (int t = 0; t < kPos; ++t) cds[t] = code_at(lane + 64 * t);
  // first matching haplotype position of every cached read position (0xFFFF: none) and whether its chain holds
  // further matches (repeats): the later passes over the votes then cost one LDS access per position
  u32 mj[kPos];
  u32 more = 0;
  // visit every (read position, matching haplotype position) pair of this lane
  auto walk = [&](auto&& fn) {
#pragma unroll
    for (int t = 0; t < kPos; ++t) {
      if (mj[t] == 0xFFFFu) continue;
      i32 const i = lane + 64 * t;
      fn(static_cast<i32>(mj[t]) - i + m);
      if (more & (1u << t)) {
        u32 const cd = cds[t];
        for (u32 j = next[mj[t]]; j != 0xFFFFu; j = next[j])
          if (same_code(code[j], cd)) fn(static_cast<i32>(j) - i + m);
      }
    }
    for (i32 i = lane + 64 * kPos; seeded && i + SK <= m; i += 64) {  // long reads: re-encode
      u32 const cd = code_at(i);
      if (cd == 0xFFFFFFFFu) continue;
      u32 const bkt = (cd * 2654435761u) >> (32 - 12);
      for (u32 j = head[bkt]; j != 0xFFFFu; j = next[j])
        if (same_code(code[j], cd)) fn(static_cast<i32>(j) - i + m);
    }
  };
  VPROF_ACC(0);
  // ---- pass A: first match of every cached read position, and is the vote unanimous? ----
  // Nine pairs in ten put every vote on ONE diagonal (error-free or nearly error-free reads, no repeat): the histogram
  // then has a single non-zero counter, so the winner is that diagonal with all the votes and the runner-up has none
  // -- no histogram update, no arg-max, no clearing.  Everything else takes the histogram path below.
  u32 nvotes = 0;  // this lane's votes; summed over the wave further down
  u32 best = 0, v2 = 0;
  i32 d2 = 0x7FFFFFFF;  // the runner-up diagonal (histogram path only; ties: the smaller)
  i32 bd = 0x7FFFFFFF;
  i32 dlo = 0x7FFFFFFF, dhi = -1;  // extreme diagonals that received a vote (histogram index: diagonal + m)
  u32 vfar8 = 0, vfar16 = 0, vfar24 = 0;  // votes further than 8 / 16 / 24 diagonals from the most-voted one
  bool unan = !(seeded && m - SK + 1 > 64 * kPos);  // long reads vote outside the cached positions
  u32 ucnt = 0;
  i32 u_d = -1;
#pragma unroll
  for (int t = 0; t < kPos; ++t) {
    u32 const cd = cds[t];
    i32 const i = lane + 64 * t;
    u32 j = 0xFFFFu;
    if (cd != 0xFFFFFFFFu) {
      j = head[(cd * 2654435761u) >> (32 - 12)];
      u32 entry = 0;
      while (j != 0xFFFFu && !same_code(entry = code[j], cd)) j = next[j];
      if (j != 0xFFFFu && (entry & kCodeDup)) more |= 1u << t;  // repeats: the chain holds further matches
    }
    mj[t] = j;
    i32 const d = j != 0xFFFFu ? static_cast<i32>(j) - i + m : -1;
    unsigned long long const have = __ballot(d >= 0);
    if (have) {
      i32 const d0 = __shfl(d, __builtin_ctzll(have));
      if (__ballot(d == d0) != have || (u_d >= 0 && d0 != u_d)) unan = false;
      u_d = d0;
      ucnt += static_cast<u32>(__popcll(have));
    }
  }
  if (__ballot(more != 0) != 0) unan = false;
  if (unan) {
    best = ucnt;
    bd = ucnt ? u_d : 0x7FFFFFFF;
    if (ucnt >= kMinChainVotes) dlo = dhi = u_d;  // one seeded diagonal: an anchor iff it holds the whole chain minimum
    if (lane == 0) nvotes = ucnt;
    VPROF_ACC(1);
    VPROF_ACC(2);
    VPROF_ACC(3);
    VPROF_ACC(4);
  } else {
    // 16-bit LDS counters: atomic add on the containing 32-bit word.  A read that sits on one diagonal makes
    // all 64 lanes hit the same counter; those same-address atomics serialise, so the first match of every
    // lane is pre-combined with a ballot and only further matches (repeats) vote one by one.
    auto vote1 = [&](i32 d) {
      atomicAdd(reinterpret_cast<u32*>(hist) + (d >> 1), 1u << ((d & 1) * 16));
      nvotes++;
    };
#pragma unroll
    for (int t = 0; t < kPos; ++t) {
      u32 const cd = cds[t];
      i32 const i = lane + 64 * t;
      u32 j = mj[t];
      i32 const d = j != 0xFFFFu ? static_cast<i32>(j) - i + m : -1;
      unsigned long long const have = __ballot(d >= 0);
      if (have) {
        i32 const d0 = __shfl(d, __builtin_ctzll(have));
        unsigned long long const same = __ballot(d == d0);
        if (same == have) {
          if (lane == __builtin_ctzll(have)) {
            u32 const cnt = static_cast<u32>(__popcll(have));
            atomicAdd(reinterpret_cast<u32*>(hist) + (d0 >> 1), cnt << ((d0 & 1) * 16));
            nvotes += cnt;
          }
        } else if (d >= 0) {
          vote1(d);
        }
      }
      if (more & (1u << t))
        for (j = next[j]; j != 0xFFFFu; j = next[j])
          if (same_code(code[j], cd)) vote1(static_cast<i32>(j) - i + m);
    }
    for (i32 i = lane + 64 * kPos; seeded && i + SK <= m; i += 64) {  // long reads
      u32 const cd = code_at(i);
      if (cd == 0xFFFFFFFFu) continue;
      for (u32 j = head[(cd * 2654435761u) >> (32 - 12)]; j != 0xFFFFu; j = next[j])
        if (same_code(code[j], cd)) vote1(static_cast<i32>(j) - i + m);
    }
    __builtin_amdgcn_wave_barrier();
    VPROF_ACC(1);
    // arg-max over the touched diagonals, ties -> smallest diagonal
    walk([&](i32 d) {
      u32 const v = hist[d];
      if (v > best || (v == best && d < bd)) {
        best = v;
        bd = d;
      }
      // anchor diagonal (oracle/align.cpp rule 2): >= kMinChainVotes votes within reach K of it
      u32 sum = v;
      if (sum < kMinChainVotes)
        for (i32 y = max(d - Kr, 0); y <= min(d + Kr, nd - 1) && sum < kMinChainVotes; ++y) sum += y != d ? hist[y] : 0u;
      if (sum >= kMinChainVotes) {
        dlo = min(dlo, d);
        dhi = max(dhi, d);
      }
    });
    for (int off = 32; off > 0; off >>= 1) {
      u32 const ob = __shfl_xor(best, off);
      i32 const od = __shfl_xor(bd, off);
      if (ob > best || (ob == best && ob > 0 && od < bd)) {
        best = ob;
        bd = od;
      }
      dlo = min(dlo, __shfl_xor(dlo, off));
      dhi = max(dhi, __shfl_xor(dhi, off));
    }
    VPROF_ACC(2);
    // second best (any other diagonal), then restore the all-zero histogram
    walk([&](i32 d) {
      if (d != bd) {
        u32 const v = hist[d];
        if (v > v2 || (v == v2 && d < d2)) {
          v2 = v;
          d2 = d;
        }
      }
      u32 const far = static_cast<u32>(d > bd ? d - bd : bd - d);  // every visit is one vote
      vfar8 += far > 8u;
      vfar16 += far > 16u;
      vfar24 += far > 24u;
    });
    for (int off = 32; off > 0; off >>= 1) {
      u32 const ov = __shfl_xor(v2, off);
      i32 const od = __shfl_xor(d2, off);
      if (ov > v2 || (ov == v2 && od < d2)) {
        v2 = ov;
        d2 = od;
      }
      vfar8 += __shfl_xor(vfar8, off);
      vfar16 += __shfl_xor(vfar16, off);
      vfar24 += __shfl_xor(vfar24, off);
    }
    __builtin_amdgcn_wave_barrier();
    VPROF_ACC(3);
    walk([&](i32 d) { hist[d] = 0; });
    __builtin_amdgcn_wave_barrier();
    VPROF_ACC(4);
  }
  if (best == 0 || dhi < 0) {  // no shared 11-mer, or only stray ones that anchor no chain: no hit
    if (lane == 0) {
      A.ws.centre[lp] = 0x7FFFFFFF;
      write_no_hit(A, lp);
    }
    return;
  }
  i32 const c = bd - m;
  // ---- search region (oracle/align.cpp rule 3) ---------------------------------
  // cost(P) = m - score(P): every read row costs >= 0 and a gap shifting the diagonal by s costs >= 12 + 3 s, so a hit
  // (score >= min_aln_score) through a seed on diagonal d stays within [d - K, d + K]: R = [vmin - K, vmax + K].
  [[maybe_unused]] i32 const ms = A.prm.min_aln_score;
  i32 const K = Kr;
  i32 const vmin = dlo - m, vmax = dhi - m;
  // ---- gapless certificates ------------------------------------------------------------------------
  // P0 = the gapless path on the most-voted diagonal c (read rows [qs, qe) against haplotype columns [rs, re)),
  // X its mismatches, S0 = (qe - qs) - 5 X.  Scores: match +1, mismatch -4, a gap of length L costs 12 + 3 L.
  // (I) and (II) show that P0 beats EVERY other overlap alignment of the two sequences, inside R or not; c is a seed
  // diagonal, so P0 lies in R and is the canonical answer.
  //
  // (I) read fully inside the haplotype (no overhang), X <= 2, no ambiguous base on P0, and no other
  //     diagonal with >= m - 10 - 11 X votes: P0 is the UNIQUE optimum:
  //   * any path with g >= 1 gaps scores <= m - 15 g < m - 5 X;
  //   * a gapless path on another diagonal d' scoring >= m - 5 X has X' <= X mismatches over >= m - 5X + 5X'
  //     bases, hence >= m - 10 - 11 X exact 11-mers, i.e. that many votes -- excluded by the vote bound.
  //
  // (II) the read overhangs ONE end of the haplotype by o bases (the overhang is soft clipped), X <= 2, no
  //     ambiguous base anywhere in read or haplotype, and 11 S0 > 6 m + 50 + 5 V_off where V_off is the number
  //     of votes that did NOT go to c.  Again P0 is the unique optimum:
  //   * a path P != P0 that pairs some row on diagonal c: rows inside the overlap gain at most 5 each, and only
  //     the X mismatch rows of P0 can gain at all; overhang rows can only be paired on diagonals c + d, d > 0,
  //     there are at most d_max of them (+1 each) and coming back to c costs a gap of total length >= d_max:
  //     score(P) - S0 <= 5 X + d_max - (12 + 3 d_max) < 0; without overhang rows P needs >= 1 gap: <= 5 X - 15 < 0.
  //   * a path that never touches c is g + 1 gapless segments on other diagonals with L_s bases and x_s
  //     mismatches; a segment contains >= L_s - 10 - 11 x_s exact 11-mers, each of which is a vote off c, so
  //     sum x_s >= (L - 10 (g+1) - V_off) / 11 with L = sum L_s <= m, and
  //     score <= L - 5 sum x_s - 15 g <= (6 m + 50 (g+1) + 5 V_off) / 11 - 15 g <= (6 m + 50 + 5 V_off) / 11 < S0.
  //   Every prefix of a unique optimum is optimal for its end cell, so the traceback's diagonal-first rule
  //   retraces P0 down to the start cell, and no other end cell reaches S0.
  // (III) NO HIT: the read overhangs one end, nothing is ambiguous, and no path INSIDE R can reach min_aln_score:
  //   * paths touching c score <= max(S0, L0 - 14) with L0 = qe - qs (first bullet of (II), any X);
  //   * paths avoiding c stay on diagonals of R; with a left overhang diagonal d pairs rows [max(0, -d), m), so all of
  //     them together pair at most Lmax = min(m, L0 + (vmax + K - c)) rows (right overhang: L0 + (c - vmin + K)),
  //     hence score <= (6 Lmax + 50 + 5 V_off) / 11 by the second bullet of (II).
  // In cases (I)/(II) (score, rs, re, qs, qe, CIGAR) is written without running the DP, in case (III) the
  // record stays "no alignment"; everything else goes to the DP kernels.
  i32 const o_left = c < 0 ? -c : 0, o_right = c + m > n ? c + m - n : 0;
  [[maybe_unused]] bool const inside = o_left == 0 && o_right == 0;
  i32 const qs = o_left, qe = m - o_right;  // overlap rows [qs, qe)
  u32 mism = 0, amb = 0, ramb = 0, namb = 0;
  if (qe - qs > 0 && (o_left == 0 || o_right == 0)) {
    for (i32 i = 32 * lane; i < m; i += 32 * 64) {  // 32 bases per lane: XOR of the bit planes
      i32 const lo = max(qs - i, 0), hi = min(qe - i, 32);  // overlap bits of this word: [lo, hi)
      u32 const rb_bad = rbad[i >> 5];
      ramb |= rb_bad;  // bits past the read's end are zero
      namb += __popc(rb_bad);
      if (hi > lo) {
        u32 const valid = (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        // haplotype bits aligned to read bit 0 of this word: haplotype position c + i + b for bit b >= lo
        i32 const hp = c + i + lo;  // >= 0
        u32 const hl = plane32(ix.hlo, hp) << lo, hh = plane32(ix.hhi, hp) << lo, hbd = plane32(ix.hbad, hp) << lo;
        u32 const bad = rb_bad | hbd;
        mism += __popc(((rlo[i >> 5] ^ hl) | (rhi[i >> 5] ^ hh) | bad) & valid);  // an ambiguous base never equals anything
        amb |= (bad & valid) ? 1u : 0u;
      }
    }
  }
  u32 vtot = nvotes;
  for (int off = 32; off > 0; off >>= 1) {
    mism += __shfl_xor(mism, off);
    amb |= __shfl_xor(amb, off);
    ramb |= __shfl_xor(ramb, off);
    vtot += __shfl_xor(vtot, off);
  }
  VPROF_ACC(5);
  if (ramb != 0) {  // (wave-uniform after the reduction; one read in a hundred)
    for (int off = m <= 256 ? 4 : 32; off > 0; off >>= 1) namb += __shfl_xor(namb, off);
    namb = static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(namb)));
  }
  i32 const X = static_cast<i32>(mism);
  i32 const v_off = static_cast<i32>(vtot) - static_cast<i32>(best);
  // ---- a ONE-GAP alignment between the two most-voted diagonals: a lower bound for the DP's optimum (vote_settle narrows the
  // region with it).  The read lies inside the haplotype on both; rows [0, t) on the first diagonal up to its first
  // mismatch, then the gap, the rest on the second: for B > A a deletion of s = B - A columns (all rows paired), for B < A an
  // insertion of s rows.  Tried both ways round -- ANY such path's score is a valid bound, the better one is taken.
  i32 lb_two = -(1 << 20);
  if (v2 > 0 && d2 != 0x7FFFFFFF && m <= 2048 && !ramb && !ix.hap_amb && !(A.prm.aln_tier & 2)) {  // (wave-uniform)
    i32 const c2 = d2 - m;
    if (c >= 0 && c + m <= n && c2 >= 0 && c2 + m <= n && c >= vmin && c <= vmax && c2 >= vmin && c2 <= vmax) {
      i32 const i = 32 * lane;
      u32 xa = 0, xb = 0;
      if (i < m) {
        u32 const valid = m - i >= 32 ? 0xFFFFFFFFu : ((1u << (m - i)) - 1u);
        u32 const rl = rlo[i >> 5], rh = rhi[i >> 5];
        xa = ((rl ^ plane32(ix.hlo, c + i)) | (rh ^ plane32(ix.hhi, c + i))) & valid;
        xb = ((rl ^ plane32(ix.hlo, c2 + i)) | (rh ^ plane32(ix.hhi, c2 + i))) & valid;
      }
      auto one_gap = [&](u32 first, u32 second, i32 a, i32 b) -> i32 {
        unsigned long long const fm = __ballot(first != 0);
        i32 t = m;
        if (fm != 0ull) {
          int const l0 = __builtin_ctzll(fm);
          t = 32 * l0 + __shfl(first ? __builtin_ctz(first) : 0, l0);
        }
        i32 const sgap = b > a ? b - a : a - b;
        i32 const start = b > a ? t : t + sgap;  // the first row on the second diagonal
        if (start > m) return -(1 << 20);
        i32 const rel = start - i;
        u32 const from = rel <= 0 ? 0xFFFFFFFFu : (rel >= 32 ? 0u : ~((1u << rel) - 1u));
        u32 xs = __popc(second & from);
        for (int off = m <= 256 ? 4 : 32; off > 0; off >>= 1) xs += __shfl_xor(xs, off);  // (words live in lanes < m / 32)
        xs = static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(xs)));
        return (b > a ? m : m - sgap) - 5 * static_cast<i32>(xs) - (GO + GE * sgap);
      };
      lb_two = max(one_gap(xa, xb, c, c2), one_gap(xb, xa, c2, c));
    }
  }
  u32 act = kVoteNoHit;
  if (lane == 0) {
    act = vote_settle(A, lp, id, m, n, c, vmin, vmax, K, X, amb != 0, ramb != 0, ix.hap_amb != 0, static_cast<i32>(v2), v_off, vfar8, vfar16, vfar24,
                      lb_two, static_cast<i32>(namb));
    if (act < kVoteFast) {  // a DP pair: act is its (width class, wall) key
      u32 const at = ix.dpbuf[64]++;
      ix.dpbuf[at] = static_cast<u32>(lp);
      ix.dpbuf[65 + at] = act;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (ix.dpbuf[64] == 64) vote_flush_dp(A, ix, lane);
  VPROF_ACC(6);
#ifdef MA_PROFILE
  if (lane == 0) {
    ix.prof[7] += 1;
    // (developer census: what keeps pairs on the general route)
    if (act < kVoteFast) {
      i32 const S0 = (qe - qs) - 5 * X;
      if (inside && !amb && X >= 3 && X <= 5 && S0 >= ms) atomicAdd(&g_vprof[11], 1ull);
      else if (inside && !amb && X <= 2) atomicAdd(&g_vprof[12], 1ull);   // vote bound failed
      else atomicAdd(&g_vprof[13], 1ull);
    }
    if (act == kVoteFast) atomicAdd(&g_vprof[10], 1ull);
    if (act == kVoteNoHit) atomicAdd(&g_vprof[9], 1ull);
  }
#endif
}

#if defined(MA_PROFILE) || defined(MA_PROFILE_TRIPS)
}  // namespace
}  // namespace ma
extern "C" void ma_debug_vprof(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ma::g_vprof), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ma::g_vprof), z, sizeof(z));
  }
}
namespace ma {
namespace {
#endif

// traceback shared by the alignment kernels (rules of oracle/align.cpp: diagonal, then E, then F;
// prefer opening a gap) + BuildCigar (genotyper.cpp:45-69).  fetch(i, t) returns the move nibble of cell (i, j),
// t = j - i - lo: bits 0-1 source of H (0 diagonal, 1 E, 2 F), bit 2 E opened here, bit 3 F opened here.
template <class Fetch>
__device__ void align_traceback(GArgs const& A, Fetch fetch, i32 lo, i32 m, i32 best, i32 bi, i32 bj, u64 gp, bool write) {
  int const MCG = A.prm.max_cigar;
  u32 ops[64];   // reversed run-length ops, len << 4 | op (private memory: only touched when a run ENDS -- the run in hand
  int nops = 0;  // lives in registers; one read-modify-write of this array per step was two memory operations per step)
  u32 total_ops = 0;
  u32 cur_op = 0xFu, cur_len = 0;
  auto flush_run = [&]() {
    if (cur_len == 0) return;
    total_ops++;
    if (nops < 64) ops[nops++] = (cur_len << 4) | cur_op; else nops = 65;  // overflow marker
    cur_len = 0;
  };
  auto push = [&](u32 op) {
    if (op != cur_op) {
      flush_run();
      cur_op = op;
    }
    cur_len++;
  };
  i32 i = bi, j = bj;
  int state = 0;
  // every step consumes a row or a column or leaves a gap state; the bound only matters if a tile were ever corrupt
  for (i32 guard = 2 * (bi + bj) + 8; guard > 0; --guard) {
    if (state == 0 && (i == 0 || j == 0)) break;
    if (i < 0 || j < 0) break;
    u32 const nib = fetch(i, j - i - lo);
    if (state == 0) {
      u32 const src = nib & 3u;
      if (src == 0) {
        push(0);
        --i;
        --j;
      } else {
        state = src == 1 ? 1 : 2;
      }
    } else if (state == 1) {
      push(2);  // D
      --j;
      if (nib & 4u) state = 0;
    } else {
      push(1);  // I
      --i;
      if (nib & 8u) state = 0;
    }
  }
  flush_run();
  if (!write) return;
  i32 const qs = i, rs = j, qe = bi, re = bj;
  u32* const acig = cig_at(A.ws, A.prm, gp);
  // BuildCigar (genotyper.cpp:45-69): S(qs) + core + S(qlen - qe); ops were collected reversed
  u32 ncig = 0, widx = 0;
  u32 o0 = 0, o1 = 0, o2 = 0, o3 = 0;  // the first four operations: inline in the record
  auto emit = [&](u32 v) {
    if (static_cast<int>(widx) < MCG) acig[1 + widx] = v;
    o0 = widx == 0 ? v : o0;
    o1 = widx == 1 ? v : o1;
    o2 = widx == 2 ? v : o2;
    o3 = widx == 3 ? v : o3;
    widx++;
    ncig++;
  };
  if (qs > 0) emit((static_cast<u32>(qs) << 4) | 4u);
  int const kept = nops > 64 ? 64 : nops;
  for (int x = kept - 1; x >= 0; --x) emit(ops[x]);
  if (qe < m) emit((static_cast<u32>(m - qe) << 4) | 4u);
  u32 const count = (nops > 64) ? (total_ops + (qs > 0) + (qe < m)) : ncig;
  acig[0] = count;
  rec_store(rec_at(A.ws, gp), best, rs, re, qs, qe, count, o0, o1, o2, o3);
}

// one DP pair of a launch: its sequences and its search region
struct DpPair {
  PairId id;
  u64 gp;  // global pair index (its record's place)
  i32 m, n, lo, wr;
  const u8* rb;
  const u8* hb;
  bool live, active;
};
__device__ __forceinline__ DpPair dp_pair_load(GArgs const& A, u32 li) {
  DpPair p{};
  p.live = li < A.dp_n;
  if (!p.live) return p;
  u64 const lp = A.ws.dp_list[A.dp0 + li];
  p.gp = A.pair0 + lp;
  {
    u32 const pr = A.ws.pair_read[lp];
    p.id.r = pr & 0x7FFFFFFu;
    p.id.slot = pr >> 27;
    p.id.w = static_cast<int>(A.ws.read_win[p.id.r]);
  }
  size_t const hi = static_cast<size_t>(p.id.w) * A.prm.max_haps + p.id.slot;
  p.n = static_cast<i32>(A.a.hap_len[hi]);
  p.hb = A.a.hap_bases + hi * A.prm.max_hap_len;
  u64 const ro = A.b.read_off[p.id.r];
  p.m = static_cast<i32>(A.b.read_off[p.id.r + 1] - ro);
  p.rb = A.b.read_bases + ro;
  p.lo = A.ws.centre[lp];
  p.wr = static_cast<i32>(A.ws.band_w[lp] & 0xFFFFu);
  p.active = p.m >= SK && p.n >= SK && static_cast<u32>(p.m) + 1 <= A.ws.tb_rows;
  return p;
}
// the record of one DP pair: no hit, or the traceback through `fetch` (see align_traceback)
template <class Fetch>
__device__ __forceinline__ void dp_pair_finish(GArgs const& A, DpPair const& p, Fetch fetch, i32 best, i32 bi, i32 bj) {
  if (!p.live) return;
  bool const hit = p.active && bi >= 0 && best >= A.prm.min_aln_score;
  if (!hit) {
    rec_at(A.ws, p.gp)[3] = 0;
    return;
  }
  align_traceback(A, fetch, p.lo, p.m, best, bi, bj, p.gp, true);
}
__device__ __forceinline__ void dp_pair_store(GArgs const& A, DpPair const& p, const u32* tb, int lane, i32 best, i32 bi,
                                              i32 bj) {
  u32 const tbw = A.ws.tb_words;
  // The walk is a chain of dependent loads, one move nibble per step (~160 steps, ~1 us each from HBM).  A step goes up a
  // row or stays in it, and leaves its 8-column word only at a gap: the words of the NEXT EIGHT ROWS at the current word
  // column are fetched together, so the chain is ~20 round trips long instead of ~160.
  constexpr int kTR = 8;
  u32 cw[kTR];
  i32 c_top = -1, c_word = -1;  // cw[r] = word c_word of row c_top - r
  dp_pair_finish(
      A, p,
      [&](i32 i, i32 t) {
        i32 const wd = t >> 3;
        if (wd != c_word || i > c_top || i <= c_top - kTR) {
          c_top = i;
          c_word = wd;
#pragma unroll
          for (int r = 0; r < kTR; ++r) cw[r] = tb[(static_cast<size_t>(max(i - r, 0)) * tbw + wd) * 64 + lane];
        }
        i32 const r = c_top - i;
        u32 wv = cw[0];
#pragma unroll
        for (int x = 1; x < kTR; ++x) wv = r == x ? cw[x] : wv;
        return (wv >> (4 * (t & 7))) & 0xFu;
      },
      best, bi, bj);
}

// ---- overlap DP over a WIDE region: one wavefront per pair ----
// Seeds spread over hundreds of diagonals (tandem repeats, segmental duplications) make regions far wider than a
// register row.  Here the 64 lanes walk a row 64 cells at a time: everything but the horizontal gap chain is
// independent per cell, and the chain E(t) = max_{t' < t} (B(t') + 3 t') - 12 - 3 t over the best non-horizontal
// entries B is one wave-wide prefix maximum plus a carry between the 64-cell chunks (the recurrence
// E(t) = max(H(t-1) - 15, E(t-1) - 3) unrolled: opening from a cell whose H is its own E never beats extending it).
// The previous row (H, F per cell) and the encoded sequences live in LDS; the move bits go to HBM as four 64-bit
// ballots per chunk.  Same cell rules, tie rules and outputs as the lane-per-pair kernels.
constexpr i32 NEGW = -(1 << 28);
// wave-wide scans with DPP (no LDS crossbar traffic: the chunk loop is one dependent chain, latency is what it costs)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ i32 dpp_mov(i32 x, i32 fill) {
  return __builtin_amdgcn_update_dpp(fill, x, CTRL, ROW_MASK, 0xF, false);
}
constexpr i32 kScanIdent = static_cast<i32>(0x80000000u);  // identity of max
__device__ __forceinline__ i32 wave_shr1(i32 x, i32 fill) { return dpp_mov<0x138, 0xF>(x, fill); }  // lane l <- lane l - 1
__device__ __forceinline__ i32 wave_excl_prefix_max(i32 x) {  // max over the lanes below this one (kScanIdent for lane 0)
  x = max(x, dpp_mov<0x111, 0xF>(x, kScanIdent));  // row_shr:1
  x = max(x, dpp_mov<0x112, 0xF>(x, kScanIdent));  // row_shr:2
  x = max(x, dpp_mov<0x114, 0xF>(x, kScanIdent));  // row_shr:4
  x = max(x, dpp_mov<0x118, 0xF>(x, kScanIdent));  // row_shr:8
  x = max(x, dpp_mov<0x142, 0xA>(x, kScanIdent));  // row_bcast:15 -> rows 1, 3
  x = max(x, dpp_mov<0x143, 0xC>(x, kScanIdent));  // row_bcast:31 -> rows 2, 3
  return wave_shr1(x, kScanIdent);
}
__global__ __launch_bounds__(64) void k_align_wave(GArgs A, u32 nchunk_alloc) {
  extern __shared__ u32 lds[];
  int const lane = threadIdx.x;
  if (blockIdx.x >= A.dp_n) return;
  // (the class's list is sorted by width step: the widest regions -- the longest workgroups -- start first)
  u32 const pi = MA_REG_ORDER ? A.dp_n - 1u - blockIdx.x : blockIdx.x;
  u64 const lp = A.ws.dp_list[A.dp0 + pi];
  u32 const pr_ = A.ws.pair_read[lp];
  PairId const id{static_cast<int>(A.ws.read_win[pr_ & 0x7FFFFFFu]), pr_ & 0x7FFFFFFu, pr_ >> 27};
  size_t const hidx = static_cast<size_t>(id.w) * A.prm.max_haps + id.slot;
  i32 const n = static_cast<i32>(A.a.hap_len[hidx]);
  const u8* hb = A.a.hap_bases + hidx * A.prm.max_hap_len;
  u64 const ro = A.b.read_off[id.r];
  i32 const m = static_cast<i32>(A.b.read_off[id.r + 1] - ro);
  const u8* rb = A.b.read_bases + ro;
  i32 const lo = A.ws.centre[lp];
  i32 const wr = static_cast<i32>(A.ws.band_w[lp] & 0xFFFFu);
  u64 const gp = A.pair0 + lp;
  bool const active = m >= SK && n >= SK && static_cast<u32>(m) + 1 <= A.ws.tb_rows;
  if (!active) {
    if (lane == 0) rec_at(A.ws, gp)[3] = 0;
    return;
  }
  u32 const GW = A.ws.gen_w;
  i32* Hrow = reinterpret_cast<i32*>(lds);          // [GW + 2] previous row, updated in place
  i32* Frow = Hrow + GW + 2;                         // [GW + 2]
  u8* hcode = reinterpret_cast<u8*>(Frow + GW + 2);  // [m + wr + 1] haplotype codes of base indices lo - 1 ... (5: outside)
  u8* qcode = hcode + ((m + wr + 4) & ~3);           // [m]
  for (i32 x = lane; x < m + wr + 1; x += 64) {
    i32 const hbidx = lo - 1 + x;
    hcode[x] = static_cast<u8>((hbidx >= 0 && hbidx < n) ? enc_base(hb[hbidx]) : 5u);
  }
  for (i32 x = lane; x < m; x += 64) qcode[x] = static_cast<u8>(enc_base(rb[x]));
  __builtin_amdgcn_wave_barrier();
  unsigned long long* tile = A.ws.tbw + static_cast<size_t>(pi) * A.ws.tb_rows * nchunk_alloc * 4;
  i32 best = NEGW, bi = -1, bj = -1;
  // overlap DP over the diagonals [blo, blo + bw): fills the tile, leaves the best end cell in (best, bi, bj)
  auto run = [&](i32 blo, i32 bw) {
    // row 0: H = 0 for 0 <= j <= n inside the band; F = minus infinity
    for (i32 t = lane; t <= bw; t += 64) {
      i32 const j = blo + t;
      Hrow[t] = (t < bw && j >= 0 && j <= n) ? 0 : NEGW;
      Frow[t] = NEGW;
    }
    __builtin_amdgcn_wave_barrier();
    i32 const nchunk = (bw + 63) >> 6;
    i32 const hshift = blo - lo;  // hcode is laid out for the full region: base index (j - 1) = lo - 1 + (i + t + hshift)
    best = NEGW;
    bi = -1;
    bj = -1;
    for (i32 i = 1; i <= m; ++i) {
      u32 const qi = qcode[i - 1];
      i32 carry = NEGW;                  // max of B(t') + 3 t' over the cells of the chunks before this one
      i32 left_h = NEGW, left_e = NEGW;  // H, E of the last cell of the chunk before this one
      for (i32 c = 0; c < nchunk; ++c) {
        i32 const t = (c << 6) + lane, j = i + blo + t;
        bool const inreg = t < bw;
        i32 const tt = inreg ? t : bw;  // clamp: cell bw is the minus-infinity sentinel
        i32 const oh = Hrow[tt], uh = Hrow[tt + (inreg ? 1 : 0)], uf = Frow[tt + (inreg ? 1 : 0)];
        u32 const hc = inreg ? hcode[i + t + hshift] : 5u;
        i32 const s = (qi > 3 || hc > 3) ? -1 : (qi == hc ? 1 : -4);
        bool const cell = inreg && j >= 1 && j <= n;  // an ordinary cell; (i, 0) is a free start, the rest does not exist
        i32 const dg = oh + s;
        i32 const fo = uh - (GO + GE), fe = uf - GE;
        i32 const f = cell ? max(fo, fe) : NEGW;
        i32 const bnh = cell ? max(dg, f) : ((inreg && j == 0) ? 0 : NEGW);  // best non-horizontal entry
        i32 const g = bnh + GE * t;
        i32 const pm = max(carry, wave_excl_prefix_max(g));
        i32 const e = cell ? pm - GO - GE * t : NEGW;
        i32 const h = max(bnh, e);
        // H, E of the left neighbour for the "gap opened here" bit
        i32 const hl = wave_shr1(h, left_h), el = wave_shr1(e, left_e);
        i32 const eo = hl - (GO + GE), ee = el - GE;
        u32 nib = (dg >= e && dg >= f) ? 0u : (e >= f ? 1u : 2u);
        nib |= (eo >= ee ? 4u : 0u) | (fo >= fe ? 8u : 0u);
        __builtin_amdgcn_wave_barrier();  // every lane has read the old row before anyone overwrites it
        if (inreg) {
          Hrow[t] = h;
          Frow[t] = f;
        }
        unsigned long long const b0 = __ballot(nib & 1u), b1 = __ballot(nib & 2u), b2 = __ballot(nib & 4u), b3 = __ballot(nib & 8u);
        if (lane < 4) tile[(static_cast<size_t>(i) * nchunk_alloc + c) * 4 + lane] = lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
        carry = max(carry, __builtin_amdgcn_readlane(max(pm, g), 63));
        left_h = __builtin_amdgcn_readlane(h, 63);
        left_e = __builtin_amdgcn_readlane(e, 63);
        // end cells: (m, j) for any j, (i, n) for i < m
        if (inreg && j >= 0 && j <= n && (i == m || j == n)) {
          if (h > best || (h == best && (i > bi || (i == bi && j < bj)))) {
            best = h;
            bi = i;
            bj = j;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    for (int off = 32; off > 0; off >>= 1) {  // best end cell of the wave: max score, ties -> larger i, then smaller j
      i32 const ob = __shfl_xor(best, off), oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
      if (ob > best || (ob == best && (oi > bi || (oi == bi && oj < bj)))) {
        best = ob;
        bi = oi;
        bj = oj;
      }
    }
  };
  // ---- narrow first pass -------------------------------------------------------------------------------------------
  // A wide region usually owes its width to a few stray anchors far from where the read really lies.  First the 64
  // diagonals [c - 32, c + 32) around the most-voted diagonal c are searched (one chunk per row).  Let C = m - score be
  // the cost of what that finds, Kc = floor((C - 12) / 3) (0 below 15) and V_far the votes further than r from c, for an
  // r <= 31 - Kc.  With nothing ambiguous in read or haplotype, ANY overlap alignment of cost <= C keeps at least
  // N = m - 10 - 11 floor(C / 5) - C mod 5 exact 11-mers (a mismatch costs 5 and breaks 11, a clipped or inserted row
  // costs >= 1 and breaks 1, a gap costs >= 15 and breaks 10), each of them a vote on the diagonal it lies on.  If
  // N > V_far one of them lies within r of c, and a path of cost <= C strays at most Kc diagonals from any diagonal it
  // touches: it stays inside [c - 31, c + 31].  So every alignment of the full region that is as good as the narrow
  // result -- the optimum and all its ties -- lies inside the narrow band, where the two searches see the same cells: same
  // end cell, same traceback.  A narrow result below min_aln_score proves nothing; the full region is searched then.
  i32 nlo = lo;
  bool settled = false;
  if (wr > 96 && (A.ws.band_w[lp] >> 17) >= static_cast<u32>(kNumReg) && !(A.prm.aln_tier & 2)) {  // (vote_aux exists for wide classes only)
    u64 const aux = A.ws.vote_aux[lp];
    i32 const c = lo + static_cast<i32>(aux & 0xFFFFu);
    u32 amb = 0;
    for (i32 x = lane; x < m + wr + 1; x += 64) amb |= hcode[x] == 4u;
    for (i32 x = lane; x < m; x += 64) amb |= qcode[x] > 3u;
    if (!__ballot(amb != 0) && c - 32 >= lo && c + 32 <= lo + wr) {
      run(c - 32, 64);
      if (bi >= 0 && best >= A.prm.min_aln_score) {
        i32 const C = m - best, Kc = C >= 15 ? (C - 12) / 3 : 0;
        i32 const rc = 31 - Kc;
        i32 const vfar = rc >= 24 ? static_cast<i32>((aux >> 48) & 0xFFFFu)
                                  : (rc >= 16 ? static_cast<i32>((aux >> 32) & 0xFFFFu) : (rc >= 8 ? static_cast<i32>((aux >> 16) & 0xFFFFu) : 0x7FFFFFFF));
        i32 const nmin = m - 10 - 11 * (C / 5) - C % 5;
        if (nmin > vfar && vfar < 65535) {
          settled = true;
          nlo = c - 32;
        }
      }
    }
  }
  if (!settled) run(lo, wr);
  bool const hit = bi >= 0 && best >= A.prm.min_aln_score;
  if (!hit) {
    if (lane == 0) rec_at(A.ws, gp)[3] = 0;
    return;
  }
  // the traceback is a serial walk; every lane follows it (uniform loads), lane 0 writes the record
  align_traceback(
      A,
      [&](i32 ri, i32 t) {
        const unsigned long long* q = tile + (static_cast<size_t>(ri) * nchunk_alloc + (t >> 6)) * 4;
        u32 const sh = static_cast<u32>(t) & 63u;
        return static_cast<u32>(((q[0] >> sh) & 1ull) | (((q[1] >> sh) & 1ull) << 1) | (((q[2] >> sh) & 1ull) << 2) |
                                (((q[3] >> sh) & 1ull) << 3));
      },
      nlo, m, best, bi, bj, gp, lane == 0);
}

// ---- last resort: a region no LDS row holds (more than ~7000 diagonals); one lane per pair, the (H,F) row in HBM ----
__global__ __launch_bounds__(64) void k_align_gen(GArgs A) {
  int const lane = threadIdx.x;
  u32 const GW = A.ws.gen_w;  // widest region of the launch: row stride
  u32* HF = A.ws.gen_row + static_cast<size_t>(blockIdx.x) * (GW + 1) * 64;  // [GW + 1][64] packed (H lo16, F hi16)
  DpPair const P = dp_pair_load(A, blockIdx.x * 64u + lane);
  i32 const m = P.m, n = P.n, lo = P.lo;
  i32 const mrows = P.active ? m : 0, wr = P.active ? P.wr : 0;
  i32 mmax = mrows, wmax = wr;  // wave-uniform loop bounds
  for (int off = 32; off > 0; off >>= 1) {
    mmax = max(mmax, __shfl_xor(mmax, off));
    wmax = max(wmax, __shfl_xor(wmax, off));
  }
  // row 0: H = 0 for 0 <= j <= n inside the region, else NEG; F = NEG.  Cells t >= wr (other lanes' wider regions, and
  // the sentinel t = wmax) stay NEG for good: nothing ever flows out of them.
  for (i32 t = 0; t <= wmax; ++t) {
    i32 const j = lo + t;
    i32 const h = (t < wr && j >= 0 && j <= n) ? 0 : NEGS;
    HF[static_cast<size_t>(t) * 64 + lane] = (static_cast<u32>(h) & 0xFFFFu) | (static_cast<u32>(NEGS) << 16);
  }
  size_t const tb_base = static_cast<size_t>(blockIdx.x) * A.ws.tb_rows * A.ws.tb_words * 64;
  u32* tb = A.ws.tb + tb_base;
  i32 best = NEGS, bi = -1, bj = -1;
  for (i32 i = 1; i <= mmax; ++i) {
    if (i <= mrows) {
      u32 const qi = enc_base(P.rb[i - 1]);
      i32 lh = NEGS, le = NEGS;  // H, E of the cell to the left (outside the region at t == 0)
      u32 word = 0;
      u32 nxt = HF[lane];  // HF[t] of the previous row, pre-loaded
      for (i32 t = 0; t < wr; ++t) {
        u32 const cur = nxt;
        nxt = HF[static_cast<size_t>(t + 1) * 64 + lane];
        i32 const dh = static_cast<i16>(cur & 0xFFFFu);
        i32 const uh = static_cast<i16>(nxt & 0xFFFFu), uf = static_cast<i16>(nxt >> 16);
        i32 const j = i + lo + t;
        u32 const tbse = (j >= 1 && j <= n) ? enc_base(P.hb[j - 1]) : 4u;
        i32 const s = (qi > 3 || tbse > 3) ? -1 : (qi == tbse ? 1 : -4);
        i32 const eo = lh - (GO + GE), ee = le - GE;
        i32 const fo = uh - (GO + GE), fe = uf - GE;
        i32 e = max(eo, ee), f = max(fo, fe);
        i32 const dg = dh + s;
        i32 h = max(dg, max(e, f));
        u32 nib = (dg >= e && dg >= f) ? 0u : (e >= f ? 1u : 2u);
        nib |= (eo >= ee ? 4u : 0u) | (fo >= fe ? 8u : 0u);
        if (j <= 0 || j > n) {
          h = (j == 0) ? 0 : NEGS;
          e = NEGS;
          f = NEGS;
        }
        // keep "minus infinity" from drifting out of i16 range
        h = max(h, 2 * NEGS + 1000);
        f = max(f, 2 * NEGS + 1000);
        e = max(e, 2 * NEGS + 1000);
        HF[static_cast<size_t>(t) * 64 + lane] = (static_cast<u32>(h) & 0xFFFFu) | (static_cast<u32>(f) << 16);
        lh = h;
        le = e;
        word |= nib << (4 * (t & 7));
        if ((t & 7) == 7 || t == wr - 1) {
          tb[(static_cast<size_t>(i) * A.ws.tb_words + (t >> 3)) * 64 + lane] = word;
          word = 0;
        }
        // end cells: (m, j) for any j, (i, n) for i < m
        if (j >= 0 && j <= n && (i == mrows || j == n)) {
          if (h > best || (h == best && (i > bi || (i == bi && j < bj)))) {
            best = h;
            bi = i;
            bj = j;
          }
        }
      }
    }
  }
  dp_pair_store(A, P, tb, lane, best, bi, bj);
}

// ---- register-resident kernel for regions of up to W diagonals ----
// The (H,F) row lives in W+1 VGPRs (packed i16x2) and the row body is fully unrolled, so the inner loop is pure VALU: no
// LDS round trip for the DP state.  The haplotype segment (4 bit/base, with wall codes 6 = "column 0", 7 = "outside the
// haplotype") stays in LDS and is re-aligned once per row with funnel shifts.  Same cell rules, tie rules and outputs as
// k_align_gen.  A pair's region may be narrower than the W cells of its class (wr in (WLO, W]): the first cell outside
// it, t = wr, is reset to "minus infinity" after every row -- a cell t only ever reads cells t and t + 1 of the row above
// and its left neighbour, so whatever the cells t > wr compute never reaches a cell of the region, and only the
// positions t in (WLO, W) carry the test.
constexpr i32 NEGR = -20000;
// DP list sorted by key = class * 2 + wall and, inside a key, by the region's width in steps of eight cells (the order never
// affects a result: every pair is independent; the DP kernels walk a row only as far as the widest region of a group)
struct KeyBase { u32 b[kNumKeys]; };
constexpr int kSubKeys = kNumKeys * 8;
__device__ __forceinline__ u32 dp_subkey(u32 bw) {  // band_w: width | key << 16
  u32 const key = bw >> 16, w = bw & 0xFFFFu;
  return key * 8u + (key < 2u * kNumReg ? min(7u, (max(w, 1u) - 1u) >> 3) : 0u);
}
// cells: (ma_timing_control mode 3 only, else null) += rows x region width of every DP pair -- the cells of the regions the
// DP is defined on (the kernels compute a class's width, in chunks of eight cells up to the widest region of a group)
__global__ __launch_bounds__(256) void k_dp_subcount(GArgs A, u32 ndp, u32* sub, unsigned long long* cells) {
  __shared__ u32 l_cnt[kSubKeys];
  __shared__ unsigned long long l_cells;
  for (u32 x = threadIdx.x; x < kSubKeys; x += 256) l_cnt[x] = 0;
  if (threadIdx.x == 0) l_cells = 0;
  __syncthreads();
  u32 const li = blockIdx.x * 256u + threadIdx.x;
  if (li < ndp) {
    u32 const lp = A.ws.dp_list[li];
    u32 const bw = A.ws.band_w[lp];
    atomicAdd(&l_cnt[dp_subkey(bw)], 1u);
    if (cells) {
      u32 const r = A.ws.pair_read[lp] & 0x7FFFFFFu;
      u64 const m = A.b.read_off[r + 1] - A.b.read_off[r];
      atomicAdd(&l_cells, static_cast<unsigned long long>(m) * (bw & 0xFFFFu));
    }
  }
  __syncthreads();
  for (u32 x = threadIdx.x; x < kSubKeys; x += 256)
    if (l_cnt[x]) atomicAdd(&sub[x], l_cnt[x]);
  if (cells && threadIdx.x == 0 && l_cells) atomicAdd(&cells[6], l_cells);
}
__global__ __launch_bounds__(256) void k_dp_scatter(GArgs A, u32 ndp, u32* out, u32* fill, const u32* sub, KeyBase kb) {
  // one global atomic per (key, width step) and WORKGROUP, all at once: the pairs take their places inside the workgroup's
  // share through LDS counters (a returning global atomic per key and wavefront, one key after the other, kept this
  // kernel waiting on a handful of contended addresses for its whole life)
  __shared__ u32 l_cnt[kSubKeys], l_base[kSubKeys], l_start[kSubKeys];
  u32 const li = blockIdx.x * 256u + threadIdx.x;
  bool const live = li < ndp;
  for (u32 x = threadIdx.x; x < kSubKeys; x += 256) l_cnt[x] = 0;
  if (threadIdx.x < kNumKeys) {  // where each width step of this key starts
    u32 at = kb.b[threadIdx.x];
    for (u32 y = 0; y < 8; ++y) {
      l_start[threadIdx.x * 8u + y] = at;
      at += sub[threadIdx.x * 8u + y];
    }
  }
  __syncthreads();
  u32 lp = 0, key = 0, pos = 0;
  if (live) {
    lp = A.ws.dp_list[li];
    key = dp_subkey(A.ws.band_w[lp]);
    pos = atomicAdd(&l_cnt[key], 1u);
  }
  __syncthreads();
  for (u32 x = threadIdx.x; x < kSubKeys; x += 256)
    if (l_cnt[x]) l_base[x] = atomicAdd(&fill[x], l_cnt[x]);
  __syncthreads();
  if (live) out[l_start[key] + l_base[key] + pos] = lp;
}

// waves per SIMD the register allocator is asked to keep (it otherwise spends registers on scheduling freedom)
#ifndef MA_RW49
#define MA_RW49 3
#endif
__host__ __device__ constexpr int reg_waves(int w) { return w <= 49 ? MA_RW49 : (w <= 65 ? 3 : (w <= 97 ? 2 : 1)); }
// the body of one width class: `group` = the class's 64-pair group this wavefront owns
template <int W, int WLO>
__device__ __forceinline__ void align_reg_body(GArgs const& A, u32 seg_words, u32 const group, u32* lds) {
  constexpr int WD = W;
  constexpr int NW = (WD + 7) / 8;  // traceback / segment words per row
  int const lane = threadIdx.x;
  u32* SEG = lds;  // [seg_words][64]
  DpPair const P = dp_pair_load(A, group * 64u + lane);
  i32 const m = P.m, n = P.n, lo = P.lo;
  i32 const wr = P.active ? P.wr : 0;
  i32 const mrows = P.active ? m : 0;
  i32 mmax = mrows, wlim = wr;
  for (int off = 32; off > 0; off >>= 1) {
    mmax = max(mmax, __shfl_xor(mmax, off));
    wlim = max(wlim, __shfl_xor(wlim, off));
  }
  wlim = __builtin_amdgcn_readfirstlane((wlim + 7) & ~7);  // rows are walked in chunks of eight cells up to the group's widest region

  // packed previous row, BIASED by what the next row subtracts anyway: lo16 = H - (GO + GE), hi16 = F - GE; HF[WD] = sentinel
  u32 HF[WD + 1];
  constexpr i32 GOE = GO + GE;
  constexpr u32 kNegPack = (static_cast<u32>(NEGR - GOE) & 0xFFFFu) | (static_cast<u32>(NEGR - GE) << 16);
  if (P.active) {
    i32 const seglen = m + WD;
    for (i32 wd = 0; wd * 8 < seglen + 8; ++wd) {
      u32 pk = 0;
      for (int x = 0; x < 8; ++x) {
        i32 const hbidx = lo + wd * 8 + x;
        u32 const e = hbidx == -1 ? 6u : ((hbidx < -1 || hbidx >= n) ? 7u : enc_base(P.hb[hbidx]));
        pk |= e << (4 * x);
      }
      if (static_cast<u32>(wd) < seg_words) SEG[static_cast<size_t>(wd) * 64 + lane] = pk;
    }
  }
#pragma unroll
  for (int t = 0; t <= WD; ++t) {
    i32 const j = lo + t;
    i32 const h = (t < wr && j >= 0 && j <= n) ? 0 : NEGR;
    HF[t] = (static_cast<u32>(h - GOE) & 0xFFFFu) | (static_cast<u32>(NEGR - GE) << 16);
  }
  size_t const tb_base = static_cast<size_t>(group) * A.ws.tb_rows * A.ws.tb_words * 64;
  u32* tb = A.ws.tb + tb_base;
  i32 best = NEGR, bi = -1, bj = -1;
  for (i32 i = 1; i <= mmax; ++i) {
    if (i <= mrows) {
      u32 const qi = enc_base(P.rb[i - 1]);
      i32 const smis = qi > 3 ? -1 : -4;
      u32 const qcmp = qi > 3 ? 15u : qi;  // never equal to a haplotype code when ambiguous
      // segment words for rel = (i-1) .. (i-1)+WD, re-aligned so that cell t uses nibble t of sw[]
      u32 const wbase = static_cast<u32>(i - 1) >> 3, sh = (static_cast<u32>(i - 1) & 7u) * 4u;
      u32 sw[NW];
      {
        u32 prev = SEG[static_cast<size_t>(wbase) * 64 + lane];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          u32 const nx = SEG[static_cast<size_t>(wbase + k + 1) * 64 + lane];
          sw[k] = sh ? ((prev >> sh) | (nx << (32u - sh))) : prev;
          prev = nx;
        }
      }
      i32 le = NEGR, last_h = NEGR;
      u32 word = 0;
      u32* tbrow = tb + static_cast<size_t>(i) * A.ws.tb_words * 64 + lane;
      // Rows whose window holds nothing but A/C/G/T for every pair of the wavefront -- no haplotype end (walls),
      // no N; almost all rows: reads hanging over a haplotype end are settled by the certificates -- take the lean
      // body: the substitution score comes from one XOR per word (nibble == 0 <=> match) and there is no wall logic.
      // Codes 4 (N), 6 and 7 (walls) all have bit 2 set, A/C/G/T do not.
      u32 any_special = 0;
#pragma unroll
      for (int k = 0; k < NW; ++k) any_special |= sw[k];
      // (the last word also carries a few columns beyond the window: including them only makes the test conservative)
      any_special &= 0x44444444u;
      i32 const smis_b = smis + GOE;  // substitution scores biased by GO + GE (see HF)
      auto const row = [&](auto special_tag) {
        constexpr bool SPECIAL = decltype(special_tag)::value;
        u32 const qrep = qcmp * 0x11111111u;
        i32 dhm = static_cast<i16>(HF[0] & 0xFFFFu);  // H(i-1, diag) - GOE; afterwards carried over from the cell before
        i32 lhm = NEGR - GOE;                          // H(i, t-1) - GOE
#pragma unroll
        for (int t0 = 0; t0 < WD; t0 += 8) {
         if (WD > 49 || t0 < wlim) {  // (the wide classes are walked whole: the chunk tests cost them registers)
#pragma unroll
        for (int t = t0; t < (t0 + 8 < WD ? t0 + 8 : WD); ++t) {
          u32 const up = HF[t + 1];
          i32 const uhm = static_cast<i16>(up & 0xFFFFu), ufm = static_cast<i32>(up) >> 16;
          i32 sb;
          u32 code = 0;
          if constexpr (SPECIAL) {
            code = (sw[t >> 3] >> (4 * (t & 7))) & 0xFu;
            sb = code > 3 ? (GOE - 1) : (code == qcmp ? (GOE + 1) : smis_b);
          } else {
            u32 const x = ((sw[t >> 3] ^ qrep) >> (4 * (t & 7))) & 0xFu;
            sb = x == 0 ? (GOE + 1) : smis_b;
          }
          i32 const dg = dhm + sb;
          i32 const eo = lhm, ee = le - GE;
          i32 const fo = uhm, fe = ufm;
          i32 e = max(eo, ee), f = max(fo, fe);
          i32 h = max(dg, max(e, f));
          u32 nib = (h == dg) ? 0u : (e >= f ? 1u : 2u);  // dg >= e && dg >= f  <=>  dg is the maximum
          nib |= (eo >= ee ? 4u : 0u) | (fo >= fe ? 8u : 0u);
          if constexpr (SPECIAL) {
            bool const wall = code >= 6;
            h = wall ? (code == 6 ? 0 : NEGR) : h;
            e = wall ? NEGR : e;
            f = wall ? NEGR : f;
            last_h = code == 7 ? last_h : h;
          } else {
            last_h = h;
          }
          i32 const hm = h - GOE, fm = f - GE;
          HF[t] = __builtin_amdgcn_perm(static_cast<u32>(fm), static_cast<u32>(hm), 0x05040100u);  // lo16(hm) | lo16(fm) << 16
          if (t > WLO) HF[t] = wr == t ? kNegPack : HF[t];  // first cell outside a narrower region
          lhm = hm;
          le = e;
          dhm = uhm;
          // nibble t & 7 of the row's word t >> 3: shifted in from the top (one v_alignbit; or-ing `nib << const` makes
          // the compiler keep 32 pre-shifted flag constants in VGPRs)
          word = __builtin_amdgcn_alignbit(nib, word, 4);
          if ((t & 7) == 7) {
            tbrow[static_cast<size_t>(t >> 3) * 64] = word;
            word = 0;
          } else if (t == WD - 1) {
            tbrow[static_cast<size_t>(t >> 3) * 64] = word >> (4 * (7 - (t & 7)));
          }
        }
         }
        }
      };
      if (__ballot(any_special != 0) != 0) row(std::true_type{}); else row(std::false_type{});
      // end cell (i, n) for i < m: the last in-haplotype cell of the row is column n iff the region reaches it
      if (i < mrows && i + lo + wr - 1 >= n && i + lo <= n) {
        if (last_h >= best) {  // later rows win ties (larger i)
          best = last_h;
          bi = i;
          bj = n;
        }
      }
    }
  }
  // end cells (m, j): scan the final row left to right (smaller j wins ties; row m beats earlier rows on ties)
  if (P.active) {
    bool first = true;
#pragma unroll
    for (int t = 0; t < WD; ++t) {
      i32 const j = mrows + lo + t;
      i32 const h = static_cast<i16>(HF[t] & 0xFFFFu) + GOE;
      if (t < wr && j >= 0 && j <= n) {
        if (h > best || (first && h == best)) {
          best = h;
          bi = mrows;
          bj = j;
        }
        if (h >= best) first = false;
      }
    }
  }
  dp_pair_store(A, P, tb, lane, best, bi, bj);
}

template <int W, int WLO>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(reg_waves(W), 8))) void k_align_reg(GArgs A, u32 seg_words) {
  extern __shared__ u32 lds[];
  align_reg_body<W, WLO>(A, seg_words, MA_REG_ORDER ? gridDim.x - 1u - blockIdx.x : blockIdx.x, lds);  // (the widest groups first)
}

// Two width classes in one launch (the first n1 workgroups are class 1's groups): a launch lasts its 150 dependent rows
// however few pairs it holds, and the classes of one batch are independent -- side by side they cost the longer of the
// two instead of their sum.  Only classes with the same register budget are paired.  A2 differs from A in the class's
// slice of the DP list, its traceback words per row and its tile base.
struct RegClass2 {
  u32 dp0, dp_n, tb_words, gen_w;
  u32* tb;
};
template <int W1, int WLO1, int W2, int WLO2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(reg_waves(W2), 8))) void k_align_reg2(GArgs A, RegClass2 c2, u32 seg_words,
                                                                                                              u32 n1) {
  extern __shared__ u32 lds[];
  static_assert(reg_waves(W1) == reg_waves(W2), "same register budget");
  if (blockIdx.x < n1) {
    align_reg_body<W1, WLO1>(A, seg_words, blockIdx.x, lds);
  } else {
    GArgs B = A;
    B.dp0 = c2.dp0;
    B.dp_n = c2.dp_n;
    B.ws.tb_words = c2.tb_words;
    B.ws.gen_w = c2.gen_w;
    B.ws.tb = c2.tb;
    align_reg_body<W2, WLO2>(B, seg_words, blockIdx.x - n1, lds);
  }
}


// ---- two pairs per lane: the register kernel's lean row body on packed 16-bit halves --------------------------------------
// A lane of align_reg_body spends ~22 vector instructions per cell on values that fit 16 bits.  Here a lane owns TWO pairs
// (lo half: pair 128 g + lane, hi half: pair 128 g + 64 + lane) and the row body runs on v_pk_add / sub / max_i16: ~24
// instructions per cell of BOTH pairs.  What has no packed form is rearranged: the four decisions of a cell (diagonal vs
// gap, E vs F, E opened, F opened) are the SIGN BITS of four packed differences, shifted into four bit planes of 16 cells
// each (plane word: lo half pair A's cells, hi half pair B's) -- the traceback reads a cell's four bits back from the planes;
// the substitution score comes from match bits (one per nibble of the XOR of read base and haplotype codes) through a packed
// multiply-add.  Only rows without walls or N in any of the wavefront's 128 haplotype segments have this form: a group that
// holds such a code anywhere falls back to align_reg_body for its two halves (the DP list is sorted by "region can reach a
// haplotype end", so those groups are few).  Cell rules, tie rules, end cells and outputs are align_reg_body's.
typedef short v2s16 __attribute__((ext_vector_type(2)));
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2u16 pk_u(u32 x) { return __builtin_bit_cast(v2u16, x); }
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return __builtin_bit_cast(u32, static_cast<v2u16>(pk_u(a) + pk_u(b))); }
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b) { return __builtin_bit_cast(u32, static_cast<v2u16>(pk_u(a) - pk_u(b))); }
__device__ __forceinline__ u32 pk_mul(u32 a, u32 b) { return __builtin_bit_cast(u32, static_cast<v2u16>(pk_u(a) * pk_u(b))); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) {
  return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(v2s16, a), __builtin_bit_cast(v2s16, b)));
}
__device__ __forceinline__ u32 pk_shr(u32 a, u32 n) {
  v2u16 const sh = {static_cast<unsigned short>(n), static_cast<unsigned short>(n)};
  return __builtin_bit_cast(u32, static_cast<v2u16>(pk_u(a) >> sh));
}
__host__ __device__ constexpr u32 pk2(i32 v) { return (static_cast<u32>(v) & 0xFFFFu) | (static_cast<u32>(v) << 16); }
__device__ __forceinline__ u32 pk_of(i32 lo16, i32 hi16) { return (static_cast<u32>(lo16) & 0xFFFFu) | (static_cast<u32>(hi16) << 16); }
__device__ __forceinline__ i32 pk_lo(u32 x) { return static_cast<i16>(x & 0xFFFFu); }
__device__ __forceinline__ i32 pk_hi(u32 x) { return static_cast<i32>(x) >> 16; }
__host__ __device__ constexpr u32 reg_pk_words(int w) { return 4u * ((static_cast<u32>(w) + 15u) / 16u); }  // plane words per row and lane

// plane rows of a 128-pair group (two 64-pair tiles side by side: row stride 2 x tb_words x 64 words)
__device__ __forceinline__ u32* pk_group_rows(GArgs const& A, u32 group) {
  return A.ws.tb + static_cast<size_t>(group) * A.ws.tb_rows * (static_cast<size_t>(2u * A.ws.tb_words) * 64);
}
template <int W, int WLO>
__device__ __forceinline__ bool align_reg_body_pk(GArgs const& A, u32 seg_words, u32 const group, u32* lds) {
  constexpr int WD = W;
  constexpr int NW = (WD + 7) / 8;    // segment words per row
  constexpr int NB = (WD + 15) / 16;  // 16-cell blocks of decision planes per row
  constexpr i32 GOE = GO + GE;
  int const lane = threadIdx.x;
  u32* const SEGA = lds;  // [seg_words][64] each
  u32* const SEGB = lds + static_cast<size_t>(seg_words) * 64;
  DpPair const PA = dp_pair_load(A, group * 128u + lane);
  DpPair const PB = dp_pair_load(A, group * 128u + 64u + lane);
  i32 const wrA = PA.active ? PA.wr : 0, wrB = PB.active ? PB.wr : 0;
  i32 const mA = PA.active ? PA.m : 0, mB = PB.active ? PB.m : 0;
  i32 mmax = max(mA, mB);
  i32 wlim = max(wrA, wrB);
  for (int off = 32; off > 0; off >>= 1) {
    mmax = max(mmax, __shfl_xor(mmax, off));
    wlim = max(wlim, __shfl_xor(wlim, off));
  }
  // Rows are walked in chunks of eight cells up to the widest region of the group's 128 pairs (the DP list is sorted by
  // width inside a class: k_dp_scatter): the regions k_vote narrows are a few cells wide, the class's row is 33 to 49.
  wlim = __builtin_amdgcn_readfirstlane((wlim + 7) & ~7);
  u32 spec = 0;
  auto const build = [&](DpPair const& P, u32* SEG) {
    if (!P.active) return;
    i32 const seglen = P.m + WD;
    for (i32 wd = 0; wd * 8 < seglen + 8; ++wd) {
      u32 pk = 0;
      for (int x = 0; x < 8; ++x) {
        i32 const hbidx = P.lo + wd * 8 + x;
        u32 const e = hbidx == -1 ? 6u : ((hbidx < -1 || hbidx >= P.n) ? 7u : enc_base(P.hb[hbidx]));
        pk |= e << (4 * x);
      }
      if (static_cast<u32>(wd) < seg_words) SEG[static_cast<size_t>(wd) * 64 + lane] = pk;
      // (only the columns k_vote's "region can reach a haplotype end" looked at -- [lo, lo + m + W + 8) -- count: the cells read
      //  none beyond lo + m + W - 2, and the last word's spare columns would send three groups in four down the general body)
      i32 const keep = seglen + 8 - wd * 8;
      spec |= keep >= 8 ? pk : (pk & ((1u << (4 * keep)) - 1u));
    }
  };
  build(PA, SEGA);
  build(PB, SEGB);
  if (__ballot((spec & 0x44444444u) != 0) != 0) return false;  // a wall or an N somewhere: the general body

  u32 H[WD + 1], F[WD + 1];  // previous row, biased as in align_reg_body: H - (GO + GE), F - GE; lo half pair A, hi half pair B
#pragma unroll
  for (int t = 0; t <= WD; ++t) {
    i32 const jA = PA.lo + t, jB = PB.lo + t;
    i32 const hA = (t < wrA && jA >= 0 && jA <= PA.n) ? 0 : NEGR;
    i32 const hB = (t < wrB && jB >= 0 && jB <= PB.n) ? 0 : NEGR;
    H[t] = pk_of(hA - GOE, hB - GOE);
    F[t] = pk2(NEGR - GE);
  }
  u32 const wrp = static_cast<u32>(wrA) | (static_cast<u32>(wrB) << 16);
  u32 const tbw = A.ws.tb_words;                              // words per row of a 64-pair tile; a group owns two such tiles
  size_t const rs = static_cast<size_t>(2u * tbw) * 64;       // row stride of the group's plane rows
  u32* const tb = pk_group_rows(A, group);
  i32 bestA = NEGR, biA = -1, bjA = -1, bestB = NEGR, biB = -1, bjB = -1;
  for (i32 i = 1; i <= mmax; ++i) {
    u32 const qA = i <= mA ? enc_base(PA.rb[i - 1]) : 4u, qB = i <= mB ? enc_base(PB.rb[i - 1]) : 4u;
    // substitution scores biased by GO + GE: mismatch (or N in the read) per half, and what a match adds to it
    i32 const smA = (qA > 3 ? -1 : -4) + GOE, smB = (qB > 3 ? -1 : -4) + GOE;
    u32 const smis = pk_of(smA, smB), delta = pk_of(GOE + 1 - smA, GOE + 1 - smB);
    u32 const qrA = (qA > 3 ? 15u : qA) * 0x11111111u, qrB = (qB > 3 ? 15u : qB) * 0x11111111u;
    // match bits of the row's cells: nibble t of zz[] holds pair A's bit (bit 0) and pair B's (bit 1)
    u32 const wbase = static_cast<u32>(i - 1) >> 3, sh = (static_cast<u32>(i - 1) & 7u) * 4u;
    u32 zz[NW];
    {
      u32 pa = SEGA[static_cast<size_t>(wbase) * 64 + lane], pb = SEGB[static_cast<size_t>(wbase) * 64 + lane];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        if (WD <= 49 && 8 * k >= wlim) break;  // (cells the row is not walked to)
        u32 const na = SEGA[static_cast<size_t>(wbase + k + 1) * 64 + lane], nb = SEGB[static_cast<size_t>(wbase + k + 1) * 64 + lane];
        u32 const xa = (sh ? ((pa >> sh) | (na << (32u - sh))) : pa) ^ qrA;
        u32 const xb = (sh ? ((pb >> sh) | (nb << (32u - sh))) : pb) ^ qrB;
        u32 const oa = xa | (xa >> 1) | (xa >> 2) | (xa >> 3), ob = xb | (xb >> 1) | (xb >> 2) | (xb >> 3);
        zz[k] = (~oa & 0x11111111u) | ((~ob & 0x11111111u) << 1);
        pa = na;
        pb = nb;
      }
    }
    u32 dhm = H[0];                  // H(i-1, diagonal) - GOE; afterwards carried over from the cell before
    u32 lhm = pk2(NEGR - GOE);       // H(i, t-1) - GOE
    u32 lem = pk2(NEGR - GE);        // E(i, t-1) - GE
    u32 acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0, hlast = 0;
    u32* const tbrow = tb + static_cast<size_t>(i) * rs + lane;
#pragma unroll
    for (int t0 = 0; t0 < WD; t0 += 8) {
     if (WD > 49 || t0 < wlim) {  // (the wide classes are walked whole: the chunk tests cost them registers)
#pragma unroll
    for (int t = t0; t < (t0 + 8 < WD ? t0 + 8 : WD); ++t) {
      u32 const uph = H[t + 1], upf = F[t + 1];
      u32 const c = (zz[t >> 3] >> (4 * (t & 7))) & 3u;
      u32 const mbit = (c * 0x8001u) & 0x00010001u;
      u32 const sb = pk_add(smis, pk_mul(mbit, delta));
      u32 const dg = pk_add(dhm, sb);
      u32 const e = pk_max(lhm, lem), f = pk_max(uph, upf);
      u32 const h = pk_max(dg, pk_max(e, f));
      // sign bits: E extended (eo < ee), F extended (fo < fe), F beats E (e < f), a gap beats the diagonal (dg < h)
      u32 const d1 = pk_sub(lhm, lem), d2 = pk_sub(uph, upf), d3 = pk_sub(e, f), d4 = pk_sub(dg, h);
      acc1 = pk_shr(acc1, 1) | (d1 & 0x80008000u);
      acc2 = pk_shr(acc2, 1) | (d2 & 0x80008000u);
      acc3 = pk_shr(acc3, 1) | (d3 & 0x80008000u);
      acc4 = pk_shr(acc4, 1) | (d4 & 0x80008000u);
      u32 hm = pk_sub(h, pk2(GOE)), fm = pk_sub(f, pk2(GE));
      lem = pk_sub(e, pk2(GE));
      if (t > WLO) {  // first cell outside a narrower region of either pair: minus infinity
        u32 const x = wrp ^ pk2(t);
        u32 const ne = __builtin_bit_cast(u32, __builtin_elementwise_min(pk_u(x), pk_u(0x00010001u)));
        u32 const msk = pk_sub(ne, 0x00010001u);  // 0xFFFF in the halves whose wr == t
        hm = (hm & ~msk) | (pk2(NEGR - GOE) & msk);
        fm = (fm & ~msk) | (pk2(NEGR - GE) & msk);
      }
      H[t] = hm;
      F[t] = fm;
      lhm = hm;
      dhm = uph;
      hlast = h;
      if ((t & 15) == 15 || t == WD - 1 || (WD <= 49 && (t & 15) == 7 && t + 1 >= wlim)) {
        u32 const fin = 15u - static_cast<u32>(t & 15);  // a partial last block: right-align its bits
        u32* const q = tbrow + static_cast<size_t>((t >> 4) * 4) * 64;
        q[0] = fin ? pk_shr(acc1, fin) : acc1;
        q[64] = fin ? pk_shr(acc2, fin) : acc2;
        q[128] = fin ? pk_shr(acc3, fin) : acc3;
        q[192] = fin ? pk_shr(acc4, fin) : acc4;
        acc1 = acc2 = acc3 = acc4 = 0;
      }
    }
     }
    }
    // end cell (i, n) for i < m: the last cell of the row is column n iff the region reaches it (see align_reg_body)
    {
      i32 const lhA = pk_lo(hlast), lhB = pk_hi(hlast);
      if (i < mA && i + PA.lo + wrA - 1 >= PA.n && i + PA.lo <= PA.n && lhA >= bestA) {
        bestA = lhA;
        biA = i;
        bjA = PA.n;
      }
      if (i < mB && i + PB.lo + wrB - 1 >= PB.n && i + PB.lo <= PB.n && lhB >= bestB) {
        bestB = lhB;
        biB = i;
        bjB = PB.n;
      }
    }
    // end cells (m, j) of a pair whose last row this is: left to right, smaller j wins ties, row m beats earlier rows on ties
    if (i == mA || i == mB) {
      bool const isA = i == mA, isB = i == mB;
      bool firstA = true, firstB = true;
#pragma unroll
      for (int t = 0; t < WD; ++t) {
        i32 const hA = pk_lo(H[t]) + GOE, hB = pk_hi(H[t]) + GOE;
        i32 const jA = i + PA.lo + t, jB = i + PB.lo + t;
        if (isA && t < wrA && jA >= 0 && jA <= PA.n) {
          if (hA > bestA || (firstA && hA == bestA)) {
            bestA = hA;
            biA = i;
            bjA = jA;
          }
          if (hA >= bestA) firstA = false;
        }
        if (isB && t < wrB && jB >= 0 && jB <= PB.n) {
          if (hB > bestB || (firstB && hB == bestB)) {
            bestB = hB;
            biB = i;
            bjB = jB;
          }
          if (hB >= bestB) firstB = false;
        }
      }
    }
  }
  // What the walk back needs goes into row 0 of the group's plane rows (the fill writes rows 1 ... m): the tracebacks are
  // chains of dependent loads, ~20 round trips a pair -- in k_align_tb2, at full occupancy, they no longer hold this
  // kernel's registers (two wavefronts per SIMD) while they wait.
  {
    u32* const q = tb + lane;
    q[0] = 1u;  // plane layout (0: the group took the general body, which walks back itself)
    q[64] = static_cast<u32>(bestA);
    q[128] = static_cast<u32>(biA);
    q[192] = static_cast<u32>(bjA);
    q[256] = static_cast<u32>(bestB);
    q[320] = static_cast<u32>(biB);
    q[384] = static_cast<u32>(bjB);
  }
  return true;
}

// k_align_reg2 with two pairs per lane.  The DP list of a class is sorted "cannot reach a haplotype end" first: those pairs go
// through the packed body in groups of 128 (segments 0 and 1: class 1, class 2), the others through align_reg_body in groups of
// 64 (segments 2 and 3) -- side by side in one launch, every workgroup one group.  (A packed group that falls back walks its two
// halves one after the other: with the wall pairs inside the packed groups, three groups in four did, and the launch lasted two
// general bodies in a row.)
struct RegSeg {
  u32 dp0, dp_n, tb_words, gen_w, units;
  u32* tb;
};
struct RegPlan { RegSeg seg[4]; };
__device__ __forceinline__ void reg_seg_apply(GArgs& A, RegSeg const& sg) {
  A.dp0 = sg.dp0;
  A.dp_n = sg.dp_n;
  A.ws.tb_words = sg.tb_words;
  A.ws.gen_w = sg.gen_w;
  A.ws.tb = sg.tb;
}
template <int W1, int WLO1, int W2, int WLO2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_align_reg2p(GArgs A, RegPlan pl, u32 seg_words) {
  extern __shared__ u32 lds[];
  static_assert(reg_waves(W1) == reg_waves(W2), "same register budget");
  // the heaviest groups first: the wider class before the narrower, packed groups (128 pairs) before the others, and -- the DP
  // list of a class is sorted by width step -- a segment's last groups before its first: the launch ends with its cheapest
  // workgroups (MA_REG_ORDER=0 at build time: in list order, as until round 6)
  u32 unit = blockIdx.x;
  int sg = 0;
  if (MA_REG_ORDER) {
    constexpr int ord[4] = {1, 0, 3, 2};
    int oi = 0;
    while (oi < 3 && unit >= pl.seg[ord[oi]].units) unit -= pl.seg[ord[oi++]].units;
    sg = ord[oi];
    unit = pl.seg[sg].units - 1u - unit;
  } else {
    while (sg < 3 && unit >= pl.seg[sg].units) unit -= pl.seg[sg++].units;
  }
  reg_seg_apply(A, pl.seg[sg]);
  if (sg == 0) {
    if (!align_reg_body_pk<W1, WLO1>(A, seg_words, unit, lds)) {
      pk_group_rows(A, unit)[threadIdx.x] = 0u;
      for (u32 hf = 0; hf < 2; ++hf) align_reg_body<W1, WLO1>(A, seg_words, 2u * unit + hf, lds);
    }
  } else if (sg == 1) {
    if (!align_reg_body_pk<W2, WLO2>(A, seg_words, unit, lds)) {
      pk_group_rows(A, unit)[threadIdx.x] = 0u;
      for (u32 hf = 0; hf < 2; ++hf) align_reg_body<W2, WLO2>(A, seg_words, 2u * unit + hf, lds);
    }
  } else if (sg == 2) {
    align_reg_body<W1, WLO1>(A, seg_words, unit, lds);
  } else {
    align_reg_body<W2, WLO2>(A, seg_words, unit, lds);
  }
}

// the walks back of k_align_reg2p's packed groups: one wavefront per 64 pairs (workgroup 2 g + h: half h of group g), a lane per pair;
// a cell's four decision bits come from the planes of its 16-cell block, the blocks of the next eight rows fetched together
__global__ __launch_bounds__(64) void k_align_tb2(GArgs A, RegPlan pl) {
  u32 g = blockIdx.x >> 1;
  u32 const half = blockIdx.x & 1u;
  int const sg = g >= pl.seg[0].units ? 1 : 0;
  if (sg) g -= pl.seg[0].units;
  reg_seg_apply(A, pl.seg[sg]);
  int const lane = threadIdx.x;
  const u32* const tb = pk_group_rows(A, g);
  if (tb[lane] != 1u) return;  // (the same for every lane of a group)
  DpPair const P = dp_pair_load(A, g * 128u + half * 64u + lane);
  size_t const rs = static_cast<size_t>(2u * A.ws.tb_words) * 64;
  i32 const best = static_cast<i32>(tb[(1 + 3 * half) * 64 + lane]), bi = static_cast<i32>(tb[(2 + 3 * half) * 64 + lane]),
            bj = static_cast<i32>(tb[(3 + 3 * half) * 64 + lane]);
  constexpr int kTR = 8;
  u32 cw[kTR][4];
  i32 c_top = -1, c_blk = -1;
  dp_pair_finish(
      A, P,
      [&](i32 ri, i32 t) {
        i32 const blk = t >> 4;
        if (blk != c_blk || ri > c_top || ri <= c_top - kTR) {
          c_top = ri;
          c_blk = blk;
#pragma unroll
          for (int r = 0; r < kTR; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) cw[r][k] = tb[static_cast<size_t>(max(ri - r, 0)) * rs + static_cast<size_t>(blk * 4 + k) * 64 + lane];
        }
        i32 const r = c_top - ri;
        u32 w1 = cw[0][0], w2 = cw[0][1], w3 = cw[0][2], w4 = cw[0][3];
#pragma unroll
        for (int x = 1; x < kTR; ++x) {
          w1 = r == x ? cw[x][0] : w1;
          w2 = r == x ? cw[x][1] : w2;
          w3 = r == x ? cw[x][2] : w3;
          w4 = r == x ? cw[x][3] : w4;
        }
        u32 const shb = half * 16u + (static_cast<u32>(t) & 15u);
        u32 const s1 = (w1 >> shb) & 1u, s2 = (w2 >> shb) & 1u, s3 = (w3 >> shb) & 1u, s4 = (w4 >> shb) & 1u;
        return (s4 ? (s3 ? 2u : 1u) : 0u) | (s1 ? 0u : 4u) | (s2 ? 0u : 8u);
      },
      best, bi, bj);
}

// ---- scoring epilogue ----
struct Cig {
  const u32* c;
  u32 n;
};
__device__ __forceinline__ u32 cop(u32 v) { return v & 0xFu; }   // 0 M, 1 I, 2 D, 4 S
__device__ __forceinline__ u32 clen(u32 v) { return v >> 4; }

// hts::ComputeEditDistance (hts/cigar_utils.h:61-111) on encoded query/target
__device__ u32 edit_distance(Cig cg, const u8* rb, u32 qn, const u8* hb, u32 tn) {
  u32 ed = 0, qp = 0, tp = 0;
  for (u32 x = 0; x < cg.n; ++x) {
    u32 const op = cop(cg.c[x]), len = clen(cg.c[x]);
    if (op == 0) {
      for (u32 y = 0; y < len; ++y, ++qp, ++tp)
        if (qp < qn && tp < tn && enc_base(rb[qp]) != enc_base(hb[tp])) ++ed;
    } else if (op == 1) {
      ed += len;
      qp += len;
    } else if (op == 2) {
      ed += len;
      tp += len;
    } else if (op == 4) {
      qp += len;
    }
  }
  return ed;
}
// hts::CigarRefPosToQueryPos (hts/cigar_utils.h:113-139)
__device__ u32 refpos_to_qpos(Cig cg, u32 ref_pos) {
  u32 qp = 0, tp = 0;
  for (u32 x = 0; x < cg.n; ++x) {
    u32 const op = cop(cg.c[x]), len = clen(cg.c[x]);
    if (op == 0) {
      for (u32 y = 0; y < len; ++y, ++qp, ++tp)
        if (tp == ref_pos) return qp;
    } else if (op == 1 || op == 4) {
      qp += len;
    } else if (op == 2) {
      for (u32 y = 0; y < len; ++y, ++tp)
        if (tp == ref_pos) return qp;
    }
  }
  return qp;
}

struct Scored {
  f64 local_score, local_identity;
  i32 global_score;
  u32 allele;
  __device__ f64 combined() const { return static_cast<f64>(global_score) + local_score * local_identity; }
};

// ScoreReadAtVariant (combined_scorer.cpp:60-108) restricted to the fields that decide the allele
__device__ Scored score_at_variant(Cig cg, i32 score, i32 rs, i32 re, const u8* rb, const u8* rq, u32 rlen,
                                   const u8* hap, i32 vstart, i32 vlen) {
  const f64* phred = reinterpret_cast<const f64*>(c_phred_bits_a);
  const i8* kMatrix = c_score_matrix;
  const u8* target = hap + rs;
  u32 const tlen = static_cast<u32>(re - rs);
  f64 pbq = 0.0, raw = 0.0;
  u32 matches = 0, aligned = 0;
  if (cg.n > 0 && vlen != 0) {  // ComputeLocalScore (local_scorer.cpp:166-279)
    i32 const vend = vstart + vlen;
    i32 tpos = 0;
    u32 qpos = 0;
    for (u32 x = 0; x < cg.n; ++x) {
      u32 const op = cop(cg.c[x]), len = clen(cg.c[x]);
      bool const cons_ref = op == 0 || op == 2;
      if (rs + tpos >= vend && cons_ref) break;
      // (only the op's positions inside [vstart, vend) do anything: they are visited directly, in the same order -- a
      //  150-base match op walked base by base for the two or three bases under a variant kept 7 of 64 lanes busy)
      i32 const y_lo = max(0, vstart - (rs + tpos));
      i32 const y_hi = min(static_cast<i32>(len), vend - (rs + tpos));
      if (op == 0) {
        for (i32 y = y_lo; y < y_hi; ++y) {
          u32 const qp = qpos + static_cast<u32>(y);
          u32 const tp = static_cast<u32>(tpos + y);
          ++aligned;
          if (!(qp >= rlen || tp >= tlen)) {
            u32 const qe = enc_base(rb[qp]), te = enc_base(target[tp]);
            i8 const r = kMatrix[te * 5 + qe];
            raw += static_cast<f64>(r);
            f64 const wgt = qp < rlen ? 1.0 - phred[rq[qp]] : 1.0;
            pbq += static_cast<f64>(r) * wgt;
            matches += (qe == te);
          }
        }
        tpos += static_cast<i32>(len);
        qpos += len;
      } else if (op == 1) {
        i32 const ap = rs + tpos;
        bool const inr = ap >= vstart && ap < vend;
        if (inr)
          for (u32 y = 0; y < len; ++y) {
            ++aligned;
            pbq += 3.0;
          }
        qpos += len;
      } else if (op == 2) {
        for (i32 y = y_lo; y < y_hi; ++y) {
          ++aligned;
          pbq += 3.0;
        }
        tpos += static_cast<i32>(len);
      } else if (op == 4) {
        qpos += len;
      }
    }
  }
  f64 const identity = aligned > 0 ? static_cast<f64>(matches) / static_cast<f64>(aligned) : 0.0;
  // ComputeSoftClipPenalty (local_scorer.cpp:290-305)
  i32 s5 = 0, s3 = 0;
  if (cg.n > 0) {
    if (cop(cg.c[0]) == 4) s5 = static_cast<i32>(clen(cg.c[0]));
    if (cg.n > 1 && cop(cg.c[cg.n - 1]) == 4) s3 = static_cast<i32>(clen(cg.c[cg.n - 1]));
  }
  f64 const sc_pen = static_cast<f64>(s5 + s3) * 4;
  f64 const global_adj = static_cast<f64>(score) - sc_pen;
  Scored s;
  s.global_score = static_cast<i32>(global_adj - raw);
  s.local_score = pbq;
  s.local_identity = identity;
  s.allele = 0;
  return s;
}

__device__ __forceinline__ u64 ev_key_of(u32 var, u32 sample, u32 allele, u32 qname) {
  return ((static_cast<u64>(var) << 44) | (static_cast<u64>(sample) << 40) | (static_cast<u64>(allele) << 33) |
          (static_cast<u64>(qname) << 1)) + 1ull;
}

// the caller's debug taps (fixed stride [read][max_haps]; cleared by the launcher): every hit's record copied to its place
__global__ __launch_bounds__(256) void k_tap_records(GArgs A, u64 total_pairs) {
  u64 const gp = static_cast<u64>(blockIdx.x) * 256 + threadIdx.x;
  if (gp >= total_pairs) return;
  const u32* r = rec_at(A.ws, gp);
  uint4 const h = *reinterpret_cast<const uint4*>(r);
  if (h.w == 0) return;
  PairId const id = pair_decode(A, gp);
  size_t const rec = static_cast<size_t>(id.r) * A.prm.max_haps + id.slot;
  if (A.o.aln_rec) {
    i32* out = A.o.aln_rec + rec * 6;
    out[0] = 1;
    out[1] = static_cast<i32>(h.x);
    out[2] = static_cast<i32>(h.y & 0xFFFFu);
    out[3] = static_cast<i32>(h.y >> 16);
    out[4] = static_cast<i32>(h.z & 0xFFFFu);
    out[5] = static_cast<i32>(h.z >> 16);
  }
  if (A.o.aln_cigar) {
    const u32* ops = h.w <= 4u ? r + 4 : cig_at(A.ws, A.prm, gp) + 1;
    u32 const nops = min(h.w, static_cast<u32>(A.prm.max_cigar));
    u32* out = A.o.aln_cigar + rec * (1 + A.prm.max_cigar);
    out[0] = h.w;
    for (u32 x = 0; x < nops; ++x) out[1 + x] = ops[x];
  }
}

// The evidence table of a window: [ev_cap] slots are reserved for every window (the largest one sizes them), a window uses
// the first ev_slots() of its own -- a power of two holding 1.5 x (its reads x its variants) keys, every read files at most
// one key per variant -- and only those are cleared per batch: 1-2 k of the 8 k slots on the whole-genome workload (the two
// whole-array memsets were 1.6 GB per step of 16384 windows, 2.3 ms of fill kernels).
__device__ __forceinline__ u32 ev_slots(u32 nrw, u32 nv, u32 cap_max) {
  u64 const need = static_cast<u64>(nrw) * nv * 3u / 2u + 16u;
  if (need >= cap_max) return cap_max;
  u32 const c = 1u << (32 - __builtin_clz(static_cast<u32>(need) - 1u));
  return min(max(c, 64u), cap_max);
}
__global__ __launch_bounds__(256) void k_ev_clear(GArgs A) {
  int const w = blockIdx.x;
  u32 const nv = A.v.win_nvars[w];
  if (nv == 0 || A.ws.win_slotmask[w] == 0) return;  // (k_assign / k_evidence leave at the same test)
  u32 const cap = ev_slots(A.b.read_win_off[w + 1] - A.b.read_win_off[w], nv, A.ws.ev_cap);
  u64* evk = A.ws.ev_key + static_cast<size_t>(w) * A.ws.ev_cap;
  u32* evm = A.ws.ev_min + static_cast<size_t>(w) * A.ws.ev_cap;
  for (u32 x = threadIdx.x; x < cap / 2; x += 256) reinterpret_cast<uint4*>(evk)[x] = make_uint4(0u, 0u, 0u, 0u);
  for (u32 x = threadIdx.x; x < cap / 4; x += 256) reinterpret_cast<uint4*>(evm)[x] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
}

// AssignReadToAlleles (genotyper.cpp:269-321): one lane per read
__global__ __launch_bounds__(64) void k_assign(GArgs A) {
  i64 const r = static_cast<i64>(blockIdx.x) * 64 + threadIdx.x;
  if (r >= A.b.n_reads) return;
  ma_params_t const& P = A.prm;
  int const MH = P.max_haps, MV = P.max_vars, MA = P.max_alts, MCG = P.max_cigar, S = P.num_samples;
  int const w = static_cast<int>(A.ws.read_win[r]);  // (k_plan)
  u8* asg = A.ws.asg_allele + static_cast<size_t>(r) * MV;
  // ("unassigned" for every variant slot: 16 bytes per store -- 64 one-byte stores per lane were most of this kernel's writes)
  auto fill255 = [&](u8* dst, int cnt) {
    if ((MV & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
      for (int v = 0; v < cnt; v += 16) *reinterpret_cast<uint4*>(dst + v) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    } else {
      for (int v = 0; v < cnt; ++v) dst[v] = 255;
    }
  };
  u32 const nv = A.v.win_nvars[w];
  // (the internal copy is read by k_evidence alone, for the window's nv variants: a window has one to three, the array 64 slots)
  fill255(asg, min(MV, static_cast<int>((nv + 15u) & ~15u)));
  if (A.o.asg_allele) fill255(A.o.asg_allele + static_cast<size_t>(r) * MV, MV);
  if (A.o.asg_score)
    for (int v = 0; v < MV; ++v) A.o.asg_score[static_cast<size_t>(r) * MV + v] = 0.0;
  u32 const mask = A.ws.win_slotmask[w];
  if (nv == 0 || mask == 0) return;
  // the records of this read: pair_off[w] + (rank of the haplotype's slot) * reads of the window + the read's place in it
  u32 const r0w = A.b.read_win_off[w], nrw = A.b.read_win_off[w + 1] - r0w;
  u64 const gp_read = A.ws.pair_off[w] + (static_cast<u64>(r) - r0w);
  u64 const ro = A.b.read_off[r];
  u32 const rlen = static_cast<u32>(A.b.read_off[r + 1] - ro);
  const u8* rb = A.b.read_bases + ro;
  const u8* rq = A.b.read_quals + ro;
  u32 sample = A.b.read_sample[r];
  u32 const qn = A.b.read_qname_id[r];
  u64* evk = A.ws.ev_key + static_cast<size_t>(w) * A.ws.ev_cap;
  u32* evm = A.ws.ev_min + static_cast<size_t>(w) * A.ws.ev_cap;
  u32 const evmask = ev_slots(nrw, nv, A.ws.ev_cap) - 1;

  for (u32 c = 0; c < A.a.win_ncomp[w]; ++c) {
    size_t const ci = static_cast<size_t>(w) * P.max_comps + c;
    u32 const hap0 = A.a.comp_hap0[ci], nh = A.a.comp_nhaps[ci];
    if (!(mask & (1u << hap0))) continue;
    for (u32 v = 0; v < nv; ++v) {
      size_t const vi = static_cast<size_t>(w) * MV + v;
      if (A.v.var_comp[vi] != c) continue;
      bool have_best = false;
      Scored bestsc{};
      for (u32 h = 0; h < nh; ++h) {  // alignments in haplotype order (all_alns)
        u64 const gp = gp_read + static_cast<u64>(__popc(mask & ((1u << (hap0 + h)) - 1u))) * nrw;
        if (!(mask & (1u << (hap0 + h)))) continue;  // (not aligned: no record)
        const u32* rp = rec_at(A.ws, gp);
        uint4 const ah = *reinterpret_cast<const uint4*>(rp);  // score, rs | re << 16, qs | qe << 16, operations
        if (!ah.w) continue;
        Cig cg{ah.w <= 4u ? rp + 4 : cig_at(A.ws, A.prm, gp) + 1, min(ah.w, static_cast<u32>(MCG))};
        // the record holds max_cigar operations; the reference scores the whole CIGAR: never silently
        if (ah.w > static_cast<u32>(MCG)) atomicOr(&A.a.win_status[w], static_cast<u32>(MA_W_CIGAR_OVERFLOW));
        // ExtractHapBounds (genotyper.cpp:329-352)
        i32 vstart, vlen;
        u32 allele;
        if (h == 0) {
          vstart = static_cast<i32>(A.v.var_ref_start[vi]);
          vlen = static_cast<i32>(A.v.var_ref_len[vi]);
          allele = 0;
        } else {
          u32 const al = A.v.var_hap_allele[vi * MH + h];
          if (al == 0) continue;  // this haplotype carries the REF allele here: no bounds
          vstart = static_cast<i32>(A.v.var_hap_start[vi * MH + h]);
          vlen = static_cast<i32>(A.v.alt_len[vi * MA + (al - 1)]);
          allele = al;
        }
        i32 const rs = static_cast<i32>(ah.y & 0xFFFFu), re = static_cast<i32>(ah.y >> 16);
        if (!((vstart + vlen) > rs && vstart < re)) continue;  // OverlapsAlignment (:360-362)
        const u8* hap = A.a.hap_bases + (static_cast<size_t>(w) * MH + hap0 + h) * P.max_hap_len;
        Scored sc = score_at_variant(cg, static_cast<i32>(ah.x), rs, re, rb, rq, rlen, hap, vstart, vlen);
        sc.allele = allele;
        if (have_best && sc.combined() <= bestsc.combined()) continue;  // first wins ties
        bestsc = sc;
        have_best = true;
      }
      if (!have_best) continue;
      asg[v] = static_cast<u8>(bestsc.allele);
      if (A.o.asg_allele) A.o.asg_allele[static_cast<size_t>(r) * MV + v] = static_cast<u8>(bestsc.allele);
      if (A.o.asg_score) A.o.asg_score[static_cast<size_t>(r) * MV + v] = bestsc.combined();
      if (sample >= static_cast<u32>(S)) continue;
      // evidence de-duplication: first read (collector order) per (variant, sample, allele, qname)
      u64 const key = ev_key_of(v, sample, bestsc.allele, qn);
      u32 slot = static_cast<u32>(key * 0x9E3779B97F4A7C15ULL >> 40) & evmask;
      for (u32 probe = 0; probe <= evmask; ++probe) {
        u64 cur = evk[slot];
        if (cur != key && cur == 0) {
          unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&evk[slot]), 0ull,
                                             static_cast<unsigned long long>(key));
          cur = old == 0ull ? key : old;
        }
        if (cur == key) {
          atomicMin(&evm[slot], static_cast<u32>(r - A.b.read_win_off[w]));
          break;
        }
        slot = (slot + 1) & evmask;
      }
    }
  }
}

__global__ __launch_bounds__(64) void k_evidence(GArgs A) {
  i64 const r = static_cast<i64>(blockIdx.x) * 64 + threadIdx.x;
  if (r >= A.b.n_reads) return;
  ma_params_t const& P = A.prm;
  int const MV = P.max_vars, NA = P.max_alts + 1, S = P.num_samples;
  int const w = static_cast<int>(A.ws.read_win[r]);
  u32 const nv = A.v.win_nvars[w];
  if (nv == 0) return;
  const u8* asg = A.ws.asg_allele + static_cast<size_t>(r) * MV;
  u32 const sample = A.b.read_sample[r];
  if (sample >= static_cast<u32>(S)) return;
  u32 const qn = A.b.read_qname_id[r];
  u32 const rev = (A.b.read_flags[r] & MA_RF_REV) ? 1u : 0u;
  const u64* evk = A.ws.ev_key + static_cast<size_t>(w) * A.ws.ev_cap;
  const u32* evm = A.ws.ev_min + static_cast<size_t>(w) * A.ws.ev_cap;
  u32 const evmask = ev_slots(A.b.read_win_off[w + 1] - A.b.read_win_off[w], nv, A.ws.ev_cap) - 1;
  u32 const rloc = static_cast<u32>(r - A.b.read_win_off[w]);
  for (u32 v = 0; v < nv; ++v) {
    u32 const al = asg[v];
    if (al == 255) continue;
    u64 const key = ev_key_of(v, sample, al, qn);
    u32 slot = static_cast<u32>(key * 0x9E3779B97F4A7C15ULL >> 40) & evmask;
    bool win = false;
    for (u32 probe = 0; probe <= evmask; ++probe) {
      u64 const cur = evk[slot];
      if (cur == key) {
        win = evm[slot] == rloc;
        break;
      }
      if (cur == 0) break;
      slot = (slot + 1) & evmask;
    }
    if (!win) continue;
    size_t const vi = static_cast<size_t>(w) * MV + v;
    atomicAdd(&A.o.allele_counts[((vi * S + sample) * NA + al) * 2 + rev], 1u);
  }
}

// Dirichlet-multinomial genotype likelihoods of one sample (caller/genotype_likelihood.cpp:93-128, :205-248): K alleles,
// K (K + 1) / 2 genotypes in VCF order; the same sequential f64 operations as the reference, lgamma from the device
// math library (the PLs are integers: a last-ulp difference in lgamma moves one only when it sits within ~1e-12 of a
// rounding boundary).  Writes the PLs (optional) and returns PL[0/0]; *gq = second-smallest PL, capped at 99.
// Round 6: every alpha is one of THREE values whatever the allele (background only / + half / + all of the main mass), so the
// 2 K lgamma calls inside every one of the 2 G genotype evaluations collapse to 3 K + 3 terms computed once -- term_j[k] is the
// very difference lgamma(c[k] + alpha) - lgamma(alpha) the reference adds up, added up in the reference's order: the same bits,
// a seventh of the calls.  KM = 5 (max_alts <= 4, the default): everything unrolled, the tables in registers -- the rolled
// version kept c[] and a frame for the inlined lgamma in scratch (214 spilled VGPRs, 692 bytes per lane).
template <int KM>
__device__ __forceinline__ u32 genotype_pls(const u32* cnt2, int K, u32* pl_out, u32* gq) {
  constexpr f64 kBackgroundError = 0.005, kOverdispersion = 0.01, kAlphaFloor = 1e-6;
  f64 const precision = (1.0 - kOverdispersion) / kOverdispersion;
  f64 const main_mass = 1.0 - kBackgroundError;
  f64 const cap = 4294967295.0 / 2.0, ln_ten = 2.302585092994045684017991454684364208;
  f64 const mu0 = kBackgroundError / K;
  f64 al[3];
  al[0] = fmax(kAlphaFloor, mu0 * precision);
  al[1] = fmax(kAlphaFloor, (mu0 + main_mass / 2.0) * precision);
  al[2] = fmax(kAlphaFloor, (mu0 + main_mass) * precision);
  f64 lg[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) lg[j] = lgamma(al[j]);
  f64 c[KM], term[3][KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    c[k] = k < K ? static_cast<f64>(cnt2[2 * k] + cnt2[2 * k + 1]) : 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) term[j][k] = k < K ? lgamma(c[k] + al[j]) - lg[j] : 0.0;
  }
  auto loglk = [&](int a, int b) {
    f64 log_prob = 0.0, alpha_sum = 0.0, count_alpha_sum = 0.0;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      if (k < K) {
        bool const hom = a == b && k == a, het = a != b && (k == a || k == b);
        f64 const alpha = hom ? al[2] : (het ? al[1] : al[0]);
        log_prob += hom ? term[2][k] : (het ? term[1][k] : term[0][k]);
        alpha_sum += alpha;
        count_alpha_sum += c[k] + alpha;
      }
    }
    return log_prob + (lgamma(alpha_sum) - lgamma(count_alpha_sum));
  };
  f64 best = -1.0e300;
  for (int b = 0; b < K; ++b)
    for (int a = 0; a <= b; ++a) best = fmax(best, loglk(a, b));
  u32 min1 = 0xFFFFFFFFu, min2 = 0xFFFFFFFFu, pl0 = 0;
  int g = 0;
  for (int b = 0; b < K; ++b)
    for (int a = 0; a <= b; ++a, ++g) {
      f64 const raw = -10.0 * (loglk(a, b) - best) / ln_ten;
      u32 const pl = static_cast<u32>(round(fmin(raw, cap)));
      if (pl_out) pl_out[g] = pl;
      if (g == 0) pl0 = pl;
      if (pl < min1) {
        min2 = min1;
        min1 = pl;
      } else if (pl < min2) {
        min2 = pl;
      }
    }
  *gq = g < 2 ? 0u : min(min2 - min1, 99u);
  return pl0;
}

// site quality + FORMAT PL / GQ (variant_call.cpp:141-163, :289-345): SOLOR over the case samples with evidence in
// case/control mode, otherwise the largest PL[0/0] over the samples with evidence
// kWithPl = false: case / control mode without the PL / GQ outputs -- the genotype likelihoods (f64 lgamma, 128 VGPRs and
// a scratch frame) are compiled out of the kernel the somatic path launches
template <bool kWithPl, int KM>
__global__ __launch_bounds__(256) void k_qual(GArgs A) {
  i64 const tix = static_cast<i64>(blockIdx.x) * blockDim.x + threadIdx.x;
  ma_params_t const& P = A.prm;
  int const MV = P.max_vars, NA = P.max_alts + 1, S = P.num_samples, G = NA * (NA + 1) / 2;
  if (tix >= static_cast<i64>(A.b.n_windows) * MV) return;
  // variant-major: a wavefront is variant slot v of 64 consecutive windows -- nearly full for the first slots, gone at once
  // for the slots no window uses (window-major put a window's two or three variants alone on a wavefront of 64 slots)
  int const w = static_cast<int>(tix % A.b.n_windows), v = static_cast<int>(tix / A.b.n_windows);
  i64 const idx = static_cast<i64>(w) * MV + v;
  // (var_qual, var_pl, var_gq are zeroed by the launcher: a thread clearing its own 30 PLs writes 120-byte strides)
  if (static_cast<u32>(v) >= A.v.win_nvars[w]) return;
  const u32* cnt = A.o.allele_counts + static_cast<size_t>(idx) * S * NA * 2;
  auto cov = [&](int s, bool alt) {
    u64 t = 0;
    for (int al = alt ? 1 : 0; al < (alt ? NA : 1); ++al) t += cnt[(s * NA + al) * 2] + cnt[(s * NA + al) * 2 + 1];
    return t;
  };
  if constexpr (kWithPl) {
    int const K = static_cast<int>(A.v.var_nalts[idx]) + 1;
    f64 qual = 0.0;
    for (int s = 0; s < S; ++s) {
      if (cov(s, false) + cov(s, true) == 0) continue;  // evidence.Find(sample) == nullptr: missing support
      u32 gq = 0;
      u32 const pl0 = genotype_pls<KM>(cnt + static_cast<size_t>(s) * NA * 2, K,
                                   A.o.var_pl ? A.o.var_pl + (static_cast<size_t>(idx) * S + s) * G : nullptr, &gq);
      if (A.o.var_gq) A.o.var_gq[static_cast<size_t>(idx) * S + s] = gq;
      qual = fmax(qual, static_cast<f64>(pl0));
    }
    if (!P.case_ctrl_mode) {
      A.o.var_qual[idx] = qual;
      return;
    }
  }
  if (!P.case_ctrl_mode) return;
  u32 const case_mask = A.ws.win_case[w];  // sample roles from the window's reads (k_plan_reads)
  f64 sum_alt = 0.0, sum_ref = 0.0, n_ctrl = 0.0;
  for (int s = 0; s < S; ++s) {
    if ((case_mask >> s) & 1u) continue;
    if (cov(s, false) + cov(s, true) == 0) continue;
    sum_alt += static_cast<f64>(cov(s, true));
    sum_ref += static_cast<f64>(cov(s, false));
    n_ctrl += 1.0;
  }
  f64 const cc = n_ctrl > 1.0 ? n_ctrl : 1.0;
  f64 const ctrl_alt = sum_alt / cc + 1.0, ctrl_ref = sum_ref / cc + 1.0;
  f64 qual = 0.0;
  for (int s = 0; s < S; ++s) {
    if (!((case_mask >> s) & 1u)) continue;
    if (cov(s, false) + cov(s, true) == 0) continue;
    f64 const case_alt = static_cast<f64>(cov(s, true)) + 1.0, case_ref = static_cast<f64>(cov(s, false)) + 1.0;
    f64 const lor = log((case_alt * ctrl_ref) / (case_ref * ctrl_alt));
    qual = qual > lor ? qual : lor;
  }
  A.o.var_qual[idx] = qual;
}

}  // namespace

int launch_genotype(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& v,
                    const ma_geno_out_t& o_in) {
  int const n = b.n_windows;
  if (n == 0) return MA_OK;
  ma_params_t const& P = ctx->prm;
  size_t const NR = static_cast<size_t>(b.n_reads);
  int const MH = P.max_haps, MV = P.max_vars, MCG = P.max_cigar;
  if (P.max_hap_len + SK > 65000 || P.max_hap_len * 2 > kIdxCap * 2) {
    ma_set_err(ctx, "max_hap_len too large for the 16-bit seed index");
    return MA_ERR_PARAM;
  }
  GArgs A{};
  A.b = b;
  A.a = a;
  A.v = v;
  A.o = o_in;
  A.prm = P;
  AlnWs& ws = A.ws;
  // evidence table: sized from the largest window (reads x a few variants each); the read count per
  // window is known from read_win_off only on the device, so a first tiny pass fetches the maxima
  ws.ev_cap = 8192;
  u32 rwords_all = 8;  // words per read bit plane (longest read of the batch + two zero words)
  {
    MA_HIP(ctx, ctx->ws_misc.reserve(4096));
    u32* cnt = ctx->ws_misc.as<u32>();
    MA_HIP(ctx, hipMemsetAsync(cnt, 0, 16, ctx->stream));
    hipLaunchKernelGGL(k_max_reads, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, b, a.win_status, cnt);
    u32 mr2[2] = {0, 0};
    MA_HIP(ctx, hipMemcpyAsync(mr2, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, ma_stream_sync(ctx));
    u32 const mr = mr2[0];
    rwords_all = (mr2[1] + 31) / 32 + 2;
    if (3u * rwords_all > 64u) {  // (cannot happen: k_max_reads leaves the windows with longer reads out)
      ma_set_err(ctx, "ma_genotype_batch: read planes do not fit a wavefront");
      return MA_ERR_PARAM;
    }
    u64 want = static_cast<u64>(mr) * 8 + 1024;
    u32 cap = 8192;
    while (cap < want && cap < (1u << 24)) cap <<= 1;
    ws.ev_cap = cap;
  }

  // fixed-size part of the workspace
  auto carve_fixed = [&](char* base) -> size_t {
    size_t off = 0;
    auto take = [&](size_t bytes) {
      off = (off + 255) & ~size_t(255);
      char* p = base ? base + off : nullptr;
      off += bytes;
      return p;
    };
    ws.win_slotmask = reinterpret_cast<u32*>(take(4ull * n));
    ws.win_case = reinterpret_cast<u32*>(take(4ull * n));
    ws.pair_off = reinterpret_cast<u64*>(take(8ull * (n + 1)));
    ws.counters = reinterpret_cast<u32*>(take(64));
    ws.vote_wg = reinterpret_cast<u32*>(take(4ull * n * MH));
    ws.read_planes = reinterpret_cast<u32*>(take(4ull * NR * plane_stride(rwords_all) + 256));
    ws.read_win = reinterpret_cast<u32*>(take(4ull * NR + 64));
    ws.ev_key = reinterpret_cast<u64*>(take(8ull * n * ws.ev_cap));
    ws.ev_min = reinterpret_cast<u32*>(take(4ull * n * ws.ev_cap));
    ws.asg_allele = reinterpret_cast<u8*>(take(NR * MV + 16));
    return off;
  };
  size_t const fixed = carve_fixed(nullptr);
  MA_HIP(ctx, ctx->ws_aln.reserve(fixed + 4096));
  carve_fixed(static_cast<char*>(ctx->ws_aln.p));
  MA_HIP(ctx, hipMemsetAsync(ws.counters, 0, 64, ctx->stream));
  MA_HIP(ctx, hipMemsetAsync(A.o.allele_counts, 0,
                             4ull * n * MV * P.num_samples * (P.max_alts + 1) * 2, ctx->stream));
  // every planned pair gets its hit flag written (a record, or an explicit "no alignment"): only the caller's debug
  // taps are cleared, so that everything that is not an alignment reads as zero there
  if (o_in.aln_rec) MA_HIP(ctx, hipMemsetAsync(A.o.aln_rec, 0, 4ull * NR * MH * 6, ctx->stream));
  if (o_in.aln_cigar) MA_HIP(ctx, hipMemsetAsync(A.o.aln_cigar, 0, 4ull * NR * MH * (1 + MCG), ctx->stream));

  ctx->tic("k_read_planes");
  if (rwords_all <= 8) hipLaunchKernelGGL(k_read_planes<8>, dim3(static_cast<u32>((NR + 31) / 32)), dim3(256), 0, ctx->stream, A, rwords_all);
  else if (rwords_all <= 16) hipLaunchKernelGGL(k_read_planes<16>, dim3(static_cast<u32>((NR + 15) / 16)), dim3(256), 0, ctx->stream, A, rwords_all);
  else hipLaunchKernelGGL(k_read_planes<32>, dim3(static_cast<u32>((NR + 7) / 8)), dim3(256), 0, ctx->stream, A, rwords_all);
  ctx->toc();
  ctx->tic("k_plan");
  hipLaunchKernelGGL(k_plan, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, A);
  hipLaunchKernelGGL(k_plan_reads, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, A);
  hipLaunchKernelGGL(k_ev_clear, dim3(n), dim3(256), 0, ctx->stream, A);  // (after k_plan: it reads the windows' slot masks)
  hipLaunchKernelGGL(k_scan_pairs, dim3(1), dim3(1024), 0, ctx->stream, ws.pair_off, n);
  ctx->toc();
  u64 total_pairs = 0;
  u32 plan_counters[4] = {0, 0, 0, 0};
  MA_HIP(ctx, hipMemcpyAsync(&total_pairs, ws.pair_off + n, 8, hipMemcpyDeviceToHost, ctx->stream));
  MA_HIP(ctx, hipMemcpyAsync(plan_counters, ws.counters, 16, hipMemcpyDeviceToHost, ctx->stream));
  MA_HIP(ctx, ma_stream_sync(ctx));
  u32 const max_read_len = plan_counters[0], n_vote_wg = plan_counters[2];

  ctx->stats[0] += total_pairs;
  // the compact alignment records: 32 bytes per planned pair (+ a side slot of 4 (1 + max_cigar) bytes that only DP pairs touch)
  MA_HIP(ctx, ctx->ws_mm.reserve((total_pairs + 64) * (32ull + 4ull * (1 + MCG)) + 512));
  ws.rec = ctx->ws_mm.as<i32>();
  ws.cig = reinterpret_cast<u32*>(reinterpret_cast<char*>(ctx->ws_mm.p) + (((total_pairs + 64) * 32ull + 255ull) & ~255ull));
  if (total_pairs > 0) {
    ws.tb_rows = max_read_len + 1;
    // LDS of k_align_wave: two i32 rows of w + 2 cells, the haplotype codes of the region and the read's codes
    auto wave_lds_bytes = [&](u32 w) { return static_cast<size_t>(w + 2) * 8 + (max_read_len + w + 8) + max_read_len + 64; };
    u32 big_w = 7168;
    while (big_w > kWaveSmallW && wave_lds_bytes(big_w) > 64 * 1024) big_w -= 64;
    if (const char* e = getenv("MA_WAVE_MAX_W")) big_w = std::max<u32>(kWaveSmallW, std::min<u32>(big_w, static_cast<u32>(atoi(e))));  // tests: reach k_align_gen
    ws.wave_big_w = big_w;
    size_t budget = std::max<size_t>(size_t(1) << 30, stage_budget(0.15, ctx->ws_misc.cap, size_t(8) << 30, ctx->hbm_share));
    if (const char* e = getenv("MA_TB_GB")) budget = static_cast<size_t>(atoi(e)) << 30;
    // vote chunks: bounded only by the 24 B / pair of region + DP lists (and 32-bit local pair ids)
    u64 const pairs_chunk = std::min<u64>(total_pairs, std::min<u64>(u64(1) << 30, budget / 4 / 28));
    size_t const tb_bytes = budget - pairs_chunk * 28;
    MA_HIP(ctx, ctx->ws_misc.reserve(std::min<size_t>(tb_bytes, std::max<size_t>(size_t(256) << 20, static_cast<size_t>((pairs_chunk + 63) / 64) *
                                                                                                    ws.tb_rows * 17 * 256)) +
                                     (pairs_chunk + 64) * 28 + 8192));
    // bytes available for traceback tiles; the per-pair arrays (and their atomics) start 256-byte aligned behind them
    size_t const tb_cap = (ctx->ws_misc.cap - ((pairs_chunk + 64) * 28 + 8192)) & ~size_t(255);
    ws.tb = ctx->ws_misc.as<u32>();
    ws.vote_aux = reinterpret_cast<u64*>(reinterpret_cast<char*>(ctx->ws_misc.p) + tb_cap);
    ws.centre = reinterpret_cast<i32*>(ws.vote_aux + pairs_chunk + 16);
    ws.pair_read = reinterpret_cast<u32*>(ws.centre) + 4 * (pairs_chunk + 16) + 640;  // behind centre / band_w / dp_list / dp_count+sorted
    ws.band_w = reinterpret_cast<u32*>(ws.centre + pairs_chunk + 16);
    ws.dp_list = ws.band_w + pairs_chunk + 16;
    ws.dp_count = ws.dp_list + pairs_chunk + 16;
    u32* const dp_sorted = ws.dp_count + 512;  // [pairs_chunk] the DP list sorted by key (k_dp_scatter); dp_count: [0, 128) counters, [128, 128 + kSubKeys) pairs per
                                               // (key, width step), [288, 288 + kSubKeys) k_dp_scatter's fill counters
    static_assert(128 + kSubKeys <= 288 && 288 + kSubKeys <= 512, "dp_count layout");
    u32 const ml_eff = std::min<u32>(static_cast<u32>(P.max_hap_len), (std::max<u32>(plan_counters[3], 64u) + 31u) & ~31u);
    u32 const hist_len = ((max_read_len + ml_eff + 2 + 1) & ~1u);
    u32 const pw_host = (ml_eff + 31) / 32 + 2;
    u32 const rwords = rwords_all;  // == (max_read_len + 31) / 32 + 2
    size_t const lds_vote = 4ull * ml_eff + 2ull * kIdxCap + 2ull * ((ml_eff + 1) & ~1) + 8ull * hist_len +
                            12ull * pw_host + 48ull * rwords + 4ull * 4 * 129 + 2ull * (ml_eff + 6) + 2ull * (plan_counters[1] + 4) +
                            8ull * ((plan_counters[1] + 3) / 4 + 10) + 64;
    if (getenv("MA_VOTE_DEBUG")) fprintf(stderr, "k_vote: %zu B of LDS (ml_eff %u, max reads %u, hist_len %u, rwords %u), %u workgroups\n", lds_vote, ml_eff, plan_counters[1], hist_len, rwords, n_vote_wg);
    if (lds_vote > 65536)
      MA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_vote), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds_vote)));
    for (u64 p0 = 0; p0 < total_pairs; p0 += pairs_chunk) {
      A.pair0 = p0;
      A.npairs = static_cast<u32>(std::min<u64>(pairs_chunk, total_pairs - p0));
      MA_HIP(ctx, hipMemsetAsync(ws.dp_count, 0, 4 * 512, ctx->stream));
      ctx->tic("k_vote");
      hipLaunchKernelGGL(k_vote, dim3(n_vote_wg), dim3(256), lds_vote, ctx->stream, A, hist_len, rwords, ml_eff);
      ctx->toc();
      u32 cnt[48];
      MA_HIP(ctx, hipMemcpyAsync(cnt, ws.dp_count, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
      MA_HIP(ctx, ma_stream_sync(ctx));
      u32 const ndp = cnt[0];
      ctx->stats[1] += ndp;
      if (ndp == 0) continue;
      if (getenv("MA_VOTE_DEBUG")) {
        fprintf(stderr, "k_vote: %u DP pairs of %u; by (class, wall):", ndp, A.npairs);
        for (int k = 0; k < kNumKeys; ++k) fprintf(stderr, " %u", cnt[4 + k]);
        fprintf(stderr, "; region widths <= 5 9 13 17 25 33 41 more (MA_DP_HIST builds):");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %u", cnt[24 + k]);
        fprintf(stderr, "\n");
      }
      GArgs const Avote = A;
      KeyBase kb{};
      u32 acc = 0;
      for (int k = 0; k < kNumKeys; ++k) {
        kb.b[k] = acc;
        acc += cnt[4 + k];
      }
      if (acc != ndp) {
        ma_set_err(ctx, "read aligner: DP pairs per width class do not add up");
        return MA_ERR_HIP;
      }
      ctx->tic("k_dp_scatter");
      unsigned long long* dstats = nullptr;
      MA_TRY_RC(ma_dev_stats(ctx, &dstats));
      hipLaunchKernelGGL(k_dp_subcount, dim3((ndp + 255) / 256), dim3(256), 0, ctx->stream, A, ndp, ws.dp_count + 128, dstats);
      hipLaunchKernelGGL(k_dp_scatter, dim3((ndp + 255) / 256), dim3(256), 0, ctx->stream, A, ndp, dp_sorted,
                         ws.dp_count + 288, ws.dp_count + 128, kb);
      ctx->toc();
      A.ws.dp_list = dp_sorted;
      auto class_n = [&](int c) { return cnt[4 + 2 * c] + cnt[4 + 2 * c + 1]; };
      auto class_w = [&](int c) -> u32 {
        return c < kNumReg ? static_cast<u32>(reg_width(c))
                           : (c == kClsWaveS ? kWaveSmallW : (c == kClsWaveB ? std::min<u32>(big_w, std::max<u32>(cnt[41], 64u)) : std::max<u32>(cnt[40], 1u)));
      };
      // A register-class launch lasts at least 150 dependent rows of W cells (0.15 - 0.6 ms for W = 33 ... 129) however few
      // pairs it holds; the wavefront kernel runs one pair per wave in ~50 us: sparse wide classes go there
      auto class_reroute = [&](int c) { return c < kNumReg && reg_width(c) >= 49 && class_n(c) <= 4096u && !getenv("MA_NO_REROUTE"); };
      auto class_wave = [&](int c) { return c == kClsWaveS || c == kClsWaveB || class_reroute(c); };
      for (int cls = 0; cls < kNumCls; ++cls) {
        u32 cls_n = class_n(cls);
        if (cls_n == 0) continue;
        ctx->stats[4 + (cls < kNumReg ? (cls < 2 ? 0 : (cls < 4 ? 1 : 2)) : 3)] += cls_n;
        u32 gw = class_w(cls);
        bool const reroute = class_reroute(cls);
        bool const wave = class_wave(cls);
        int last = cls;
        if (wave) {
          // the wavefront kernel takes every pair's own width from its record: the classes it serves that follow each other
          // in the sorted list (nothing of a register class in between) share ONE launch, sized for the widest of them --
          // each launch lasts a pair's 150+ dependent rows however few pairs it holds
          for (int c2 = cls + 1; c2 < kNumCls; ++c2) {
            u32 const n2 = class_n(c2);
            if (n2 == 0) continue;
            if (!class_wave(c2)) break;
            ctx->stats[4 + (c2 < kNumReg ? (c2 < 2 ? 0 : (c2 < 4 ? 1 : 2)) : 3)] += n2;
            cls_n += n2;
            gw = std::max(gw, class_w(c2));
            last = c2;
          }
        }
        // two register classes of the same register budget, side by side in one launch (k_align_reg2) when both fit the
        // traceback workspace whole
        // (only the classes of at most 49 cells have a paired / packed instantiation: class 3 -- 65 cells -- shares
        //  reg_waves() == 3 with them when MA_RW49 is 3 and must NOT be taken for one of them)
        auto pair_class = [&](int c) { return c < kNumReg && reg_width(c) <= 49; };
        if (!wave && pair_class(cls) && !getenv("MA_NO_PAIR")) {
          int c2 = -1;
          for (int c = cls + 1; c < kNumReg; ++c) {
            if (class_n(c) == 0) continue;
            if (!class_wave(c) && pair_class(c)) c2 = c;
            break;
          }
          // (all three 49-register classes in use -- the narrowed regions fill the first: it runs alone, the other two side by
          //  side, two full launches instead of a full one and a sparse one)
          if (cls == 0 && c2 == 1 && class_n(2) != 0 && !class_wave(2)) c2 = -1;
          // (a class without a partner still takes the packed launch, beside an EMPTY neighbour class: with the regions
          //  narrowed there are pairs in all three classes, and the third fell back to one pair per lane)
          bool const pk_env = !(getenv("MA_ALIGN_PK") && atoi(getenv("MA_ALIGN_PK")) == 0);
          bool const solo = c2 < 0 && pk_env;
          int const lo_c = solo ? (cls == 0 ? 0 : cls - 1) : cls, hi_c = solo ? (cls == 0 ? 1 : cls) : c2;
          if (c2 >= 0 || solo) {
            u32 const n1 = cls_n, n2 = solo ? 0u : class_n(c2);
            u32 const gw2 = solo ? class_w(hi_c) : class_w(c2);
            // (36 % fewer vector instructions.  Round 4 left it opt-in: the four-lane step was no shorter.  Round 5's step is bound
            //  by the throughput kernels' CU time, and the packed launch now buys +1.7 % (A/B on one box, twice); MA_ALIGN_PK=0
            //  is the one-pair-per-lane launch)
            bool const pk = pk_env;
            if (pk) {
              // two pairs per lane (k_align_reg2p): the pairs that cannot reach a haplotype end in packed groups of 128 (two
              // 64-pair tiles, wide enough for the plane rows), the others in groups of 64 through the general body
              u32 const gwl = solo ? class_w(lo_c) : gw;
              u32 const tw[2] = {std::max<u32>((gwl + 7) / 8, reg_pk_words(static_cast<int>(gwl)) / 2),
                                 std::max<u32>((gw2 + 7) / 8, reg_pk_words(static_cast<int>(gw2)) / 2)};
              u32 const gws[2] = {gwl, gw2};
              int const cc[2] = {lo_c, hi_c};
              bool const live[2] = {!solo || lo_c == cls, !solo || hi_c == cls};
              RegPlan pl{};
              size_t off = 0;  // bytes into the traceback workspace
              for (int sgi = 0; sgi < 4; ++sgi) {
                int const x = sgi & 1;
                bool const packed = sgi < 2;
                u32 const n = live[x] ? cnt[4 + 2 * cc[x] + (packed ? 0 : 1)] : 0u;
                u32 const per = packed ? 128u : 64u;
                RegSeg& sg = pl.seg[sgi];
                sg.dp0 = kb.b[2 * cc[x] + (packed ? 0 : 1)];
                sg.dp_n = n;
                sg.tb_words = tw[x];
                sg.gen_w = gws[x];
                sg.units = (n + per - 1) / per;
                sg.tb = A.ws.tb + off / 4;
                off += static_cast<size_t>(sg.units) * ws.tb_rows * tw[x] * per * 4;
              }
              u32 const units = pl.seg[0].units + pl.seg[1].units + pl.seg[2].units + pl.seg[3].units;
              if (hi_c > 2 || lo_c >= hi_c || gwl > static_cast<u32>(reg_width(lo_c)) || gw2 > static_cast<u32>(reg_width(hi_c))) {
                ma_set_err(ctx, "read aligner: a width class without a packed instantiation reached the packed launch");
                return MA_ERR_HIP;
              }
              if (off <= tb_cap && units > 0) {
                if (!solo) ctx->stats[4 + (c2 < 2 ? 0 : (c2 < 4 ? 1 : 2))] += n2;
                u32 const segw = (max_read_len + gw2 + 7) / 8 + 3;
                size_t const lds_reg = static_cast<size_t>(segw) * 512;
                ctx->tic("k_align_reg");
                if (lo_c == 0 && hi_c == 1)
                  hipLaunchKernelGGL((k_align_reg2p<reg_width(0), 0, reg_width(1), reg_width(0)>), dim3(units), dim3(64), lds_reg, ctx->stream,
                                     A, pl, segw);
                else if (lo_c == 0 && hi_c == 2)
                  hipLaunchKernelGGL((k_align_reg2p<reg_width(0), 0, reg_width(2), reg_width(1)>), dim3(units), dim3(64), lds_reg, ctx->stream,
                                     A, pl, segw);
                else
                  hipLaunchKernelGGL((k_align_reg2p<reg_width(1), reg_width(0), reg_width(2), reg_width(1)>), dim3(units), dim3(64), lds_reg,
                                     ctx->stream, A, pl, segw);
                ctx->toc();
                if (pl.seg[0].units + pl.seg[1].units > 0) {
                  ctx->tic("k_align_tb");
                  hipLaunchKernelGGL(k_align_tb2, dim3(2 * (pl.seg[0].units + pl.seg[1].units)), dim3(64), 0, ctx->stream, A, pl);
                  ctx->toc();
                }
                if (!solo) cls = c2;
                continue;
              }
            }
            if (!solo) {  // (a lone class without room for the packed tiles: one pair per lane below)
            u32 const tw1 = (gw + 7) / 8, tw2 = (gw2 + 7) / 8;
            size_t const tpg1 = static_cast<size_t>(ws.tb_rows) * tw1 * 64 * 4, tpg2 = static_cast<size_t>(ws.tb_rows) * tw2 * 64 * 4;
            u32 const ng1 = (n1 + 63) / 64, ng2 = (n2 + 63) / 64;
            if (static_cast<size_t>(ng1) * tpg1 + static_cast<size_t>(ng2) * tpg2 <= tb_cap) {
              ctx->stats[4 + (c2 < 2 ? 0 : (c2 < 4 ? 1 : 2))] += n2;
              A.ws.tb_words = tw1;
              A.ws.gen_w = gw;
              A.dp0 = kb.b[2 * cls];
              A.dp_n = n1;
              RegClass2 const second{kb.b[2 * c2], n2, tw2, gw2, A.ws.tb + static_cast<size_t>(ng1) * tpg1 / 4};
              u32 const segw = (max_read_len + gw2 + 7) / 8 + 3;
              size_t const lds_reg = static_cast<size_t>(segw) * 256;
              ctx->tic("k_align_reg");
              if (cls == 0 && c2 == 1)
                hipLaunchKernelGGL((k_align_reg2<reg_width(0), 0, reg_width(1), reg_width(0)>), dim3(ng1 + ng2), dim3(64), lds_reg,
                                   ctx->stream, A, second, segw, ng1);
              else if (cls == 0 && c2 == 2)
                hipLaunchKernelGGL((k_align_reg2<reg_width(0), 0, reg_width(2), reg_width(1)>), dim3(ng1 + ng2), dim3(64), lds_reg,
                                   ctx->stream, A, second, segw, ng1);
              else
                hipLaunchKernelGGL((k_align_reg2<reg_width(1), reg_width(0), reg_width(2), reg_width(1)>), dim3(ng1 + ng2), dim3(64),
                                   lds_reg, ctx->stream, A, second, segw, ng1);
              ctx->toc();
              cls = c2;
              continue;
            }
            }
          }
        }
        A.ws.tb_words = (gw + 7) / 8;
        A.ws.gen_w = gw;
        u32 const nchunk = (gw + 63) / 64;
        // bytes of traceback per launch group: 64 pairs of a lane-per-pair kernel, ONE pair of the wavefront kernel
        size_t const tb_per_group = wave ? static_cast<size_t>(ws.tb_rows) * nchunk * 32
                                         : static_cast<size_t>(ws.tb_rows) * A.ws.tb_words * 64 * 4;
        u64 groups_max = std::max<u64>(1, tb_cap / tb_per_group);
        A.ws.tbw = reinterpret_cast<unsigned long long*>(ws.tb);
        if (cls == kClsGlobal) {  // (H,F) rows in HBM: one row of gw + 1 words per lane
          size_t const row_bytes = (static_cast<size_t>(gw) + 1) * 256;
          groups_max = std::min<u64>(groups_max, std::max<u64>(1, (size_t(2) << 30) / row_bytes));
          MA_HIP(ctx, ctx->ws_gen.reserve(std::min<u64>(groups_max, (cls_n + 63) / 64) * row_bytes));
          A.ws.gen_row = ctx->ws_gen.as<u32>();
        }
        if (tb_per_group > tb_cap) {
          ma_set_err(ctx, "read aligner: traceback tile of one pair group exceeds the workspace");
          return MA_ERR_NOMEM;
        }
        u32 const per_group = wave ? 1u : 64u;  // pairs per workgroup
        u64 const ng_total = (static_cast<u64>(cls_n) + per_group - 1) / per_group;
        for (u64 g0 = 0; g0 < ng_total; g0 += groups_max) {
          u32 const ng = static_cast<u32>(std::min<u64>(groups_max, ng_total - g0));
          A.dp0 = kb.b[2 * cls] + static_cast<u32>(g0 * per_group);
          A.dp_n = static_cast<u32>(std::min<u64>(static_cast<u64>(ng) * per_group, cls_n - g0 * per_group));
          u32 const segw = (max_read_len + gw + 7) / 8 + 3;
          size_t const lds_reg = static_cast<size_t>(segw) * 256;
#define MA_LAUNCH_REG(C, WLO)                                                                                          \
  case C:                                                                                                              \
    ctx->tic("k_align_reg");                                                                                           \
    hipLaunchKernelGGL((k_align_reg<reg_width(C), WLO>), dim3(ng), dim3(64), lds_reg, ctx->stream, A, segw);          \
    ctx->toc();                                                                                                        \
    break;
          switch ((reroute || wave) ? kClsWaveB : cls) {
            MA_LAUNCH_REG(0, 0)
            MA_LAUNCH_REG(1, reg_width(0))
            MA_LAUNCH_REG(2, reg_width(1))
            MA_LAUNCH_REG(3, reg_width(2))
            MA_LAUNCH_REG(4, reg_width(3))
            MA_LAUNCH_REG(5, reg_width(4))
            case kClsWaveS:
            case kClsWaveB: {
              size_t const lds_w = wave_lds_bytes(gw);
              ctx->tic("k_align_wave");
              hipLaunchKernelGGL(k_align_wave, dim3(ng), dim3(64), lds_w, ctx->stream, A, nchunk);
              ctx->toc();
              break;
            }
            default:
              ctx->tic("k_align_gen");
              hipLaunchKernelGGL(k_align_gen, dim3(ng), dim3(64), 0, ctx->stream, A);
              ctx->toc();
              break;
          }
#undef MA_LAUNCH_REG
        }
        cls = last;  // (the classes the launch swallowed)
      }
      A = Avote;
    }
  }
  if ((o_in.aln_rec || o_in.aln_cigar) && total_pairs > 0) {
    ctx->tic("k_tap_records");
    hipLaunchKernelGGL(k_tap_records, dim3(static_cast<u32>((total_pairs + 255) / 256)), dim3(256), 0, ctx->stream, A, total_pairs);
    ctx->toc();
  }
  ctx->tic("k_assign");
  hipLaunchKernelGGL(k_assign, dim3(static_cast<u32>((NR + 63) / 64)), dim3(64), 0, ctx->stream, A);
  ctx->toc();
  ctx->tic("k_evidence");
  hipLaunchKernelGGL(k_evidence, dim3(static_cast<u32>((NR + 63) / 64)), dim3(64), 0, ctx->stream, A);
  ctx->toc();
  ctx->tic("k_qual");
  {
    int const NAq = P.max_alts + 1, Gq = NAq * (NAq + 1) / 2;
    MA_HIP(ctx, hipMemsetAsync(A.o.var_qual, 0, 8ull * n * MV, ctx->stream));
    if (A.o.var_pl) MA_HIP(ctx, hipMemsetAsync(A.o.var_pl, 0, 4ull * n * MV * P.num_samples * Gq, ctx->stream));
    if (A.o.var_gq) MA_HIP(ctx, hipMemsetAsync(A.o.var_gq, 0, 4ull * n * MV * P.num_samples, ctx->stream));
    bool const with_pl = !P.case_ctrl_mode || A.o.var_pl || A.o.var_gq;
    auto kq = with_pl ? (NAq <= 5 ? k_qual<true, 5> : k_qual<true, 16>) : k_qual<false, 5>;
    hipLaunchKernelGGL(kq, dim3(static_cast<u32>((static_cast<size_t>(n) * MV + 255) / 256)), dim3(256), 0, ctx->stream, A);
  }
  ctx->toc();
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

}  // namespace ma
