// Variant annotation: SEQ_CX (11 sequence-complexity features) and GRAPH_CX per variant.
//   core/variant_annotator.cpp:43-101      AnnotateSequenceComplexity / AnnotateGraphComplexity
//   base/sequence_complexity.cpp:31-507    flanks, homopolymer run, Shannon entropy, STR finder, MergeMax
//   base/longdust_scorer.h:217-462         LongdustQ k-mer concentration score (k=4 flanks, k=7 haplotype)
//
// Two kernels, both tiny next to the aligners (one pass over <= 16 haplotypes of <= 4 kb per window):
//   k_hap_lq  one workgroup per (window, component): LongdustQ(k=7) of the component's REF haplotype, 4^7 count
//             table in LDS (64 KB), both strands from the same table (count_rc[x] = count_fwd[rc(x)]).
//   k_seqcx   one wavefront per variant: the +-50 window of every (ALT, haplotype) site staged in LDS; lanes
//             cooperate on the counting passes and evaluate 64 candidate repeat starts at a time.
// Exactness: every integer feature and every f32/f64 add, multiply and divide follows the reference's operation
// ORDER (the log-factorial sums run in k-mer index order over a host-built lgamma table, the null-model table
// f(l) is built on the host with the same series), so the only device-libm calls are log2f (entropy), log1p
// (LongdustQ squash) and log10 (GEI): parity is bit-exact for the integers and <= 1e-5 for the floats.
#include <algorithm>
#include <cmath>
#include <vector>

#include "ma_internal.h"

namespace ma {
namespace {

// ---- host: null-model table f(l) (longdust_scorer.h:350-451) and log-factorial table -------------------
f64 poisson_log_factorial(f64 lambda) {  // E[log(X!)] for X ~ Poisson(lambda), longdust_scorer.h:350-389
  if (lambda < 1e-10) return 0.0;
  if (lambda >= 30.0) {
    f64 const inv = 1.0 / lambda;
    f64 const pi = 3.141592653589793238462643383279502884, e = 2.718281828459045235360287471352662498;
    f64 const stirling =
        (0.5 * std::log(2.0 * pi * e * lambda)) - (inv / 12.0 * (1.0 + (0.5 * inv) + (19.0 / 30.0 * inv * inv)));
    return stirling + (lambda * (std::log(lambda) - 1.0));
  }
  f64 accum = 0.0, log_fact = 0.0, term = lambda;
  for (int c = 2; c <= 10000; ++c) {
    log_fact += std::log(static_cast<f64>(c));
    term *= lambda / c;
    f64 const z = term * log_fact;
    if (z < accum * 1e-9) break;
    accum += z;
  }
  return accum * std::exp(-lambda);
}

f64 null_model_f(int k, f64 gc, int ell) {  // longdust_scorer.h:399-440
  u32 const num_kmers = 1u << (2 * k);
  if (std::abs(gc - 0.5) < 1e-6) {
    f64 const lambda = static_cast<f64>(ell) / num_kmers;
    return static_cast<f64>(num_kmers) * poisson_log_factorial(lambda);
  }
  f64 const safe_gc = std::clamp(gc, 1e-6, 1.0 - 1e-6);
  f64 const p_gc = safe_gc / 2.0, p_at = (1.0 - safe_gc) / 2.0;
  f64 const two_pow_k = static_cast<f64>(1ULL << k);
  f64 total = 0.0;
  for (int g = 0; g <= k; ++g) {
    f64 comb = 1.0;
    for (int j = 1; j <= g; ++j) comb *= static_cast<f64>(k - j + 1) / static_cast<f64>(j);
    f64 const n_kmers = comb * two_pow_k;
    f64 const prob = std::pow(p_gc, g) * std::pow(p_at, k - g);
    total += n_kmers * poisson_log_factorial(static_cast<f64>(ell) * prob);
  }
  return total;
}

struct CxTables {
  const f64* f4;    // [ml + 1] f(l) at k = 4
  const f64* f7;    // [ml + 1] f(l) at k = 7
  const f64* lgam;  // [ml + 2] lgamma(c + 1)
};

struct CxArgs {
  int n_windows;
  int MC, MH, ML, MV, MA;
  ma_asm_out_t a;
  ma_var_out_t v;
  ma_cx_out_t o;
  CxTables t;
  f64* comp_hlq;  // [n * MC] log1p(LongdustQ k=7 of the component's REF haplotype)
};

// ---- device helpers ----------------------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ f64 read_lane_f64(f64 v, u32 lane) {
  u64 const b = __double_as_longlong(v);
  u32 const lo = __builtin_amdgcn_readlane(static_cast<u32>(b), lane);
  u32 const hi = __builtin_amdgcn_readlane(static_cast<u32>(b >> 32), lane);
  return __longlong_as_double((static_cast<u64>(hi) << 32) | lo);
}

__device__ __forceinline__ i32 wave_max_i32(i32 v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = max(v, __shfl_xor(v, s, 64));
  return v;
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// reverse complement of a k-mer code (2 bits per base, first base in the high bits)
__device__ __forceinline__ u32 rc_code(u32 x, int k) {
  u32 y = __brev(~x);                                       // bit-reversed complement: base order reversed, bits swapped
  y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);  // swap the two bits of every base back
  return y >> (32 - 2 * k);
}

// LongdustQScorer::ScoreOneStrand epilogue (longdust_scorer.h:311-329)
__device__ __forceinline__ f64 q_of(f64 sum_log_fact, u32 valid, const f64* ftab) {
  f64 const q = sum_log_fact - ftab[valid];
  return fmax(0.0, q / static_cast<f64>(valid));
}

// ---- k_hap_lq: LongdustQ(k=7) of each component's REF haplotype ----------------------------------------
constexpr int kHapK = 7;
constexpr u32 kHapBins = 1u << (2 * kHapK);
constexpr int kHlqThreads = 256;
constexpr u32 kListCap = 2048;  // k-mers with count >= 2 of a haplotype of <= 4096 bases
// one pad word per 256 bins: the reverse-strand pass reads bins 256 apart across the lanes of a wavefront
// (rc() moves the last three bases to the front), which would be a 64-way LDS bank conflict unpadded
__device__ __forceinline__ u32 padded(u32 bin) { return bin + (bin >> 8); }

__global__ __launch_bounds__(kHlqThreads) void k_hap_lq(CxArgs A) {
  __shared__ u32 cnt[kHapBins + (kHapBins >> 8)];
  __shared__ f64 list[2][kListCap];
  __shared__ u32 wave_tot[2][4];
  __shared__ u32 n_valid;
  int const w = blockIdx.x / A.MC, c = blockIdx.x % A.MC;
  u32 const tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  size_t const ci = static_cast<size_t>(w) * A.MC + c;
  if (A.v.win_nvars[w] == 0 || static_cast<u32>(c) >= A.a.win_ncomp[w]) return;
  size_t const hi = static_cast<size_t>(w) * A.MH + A.a.comp_hap0[ci];
  u32 const L = A.a.hap_len[hi];
  const u8* s = A.a.hap_bases + hi * A.ML;
  for (u32 i = tid; i < kHapBins + (kHapBins >> 8); i += kHlqThreads) cnt[i] = 0;
  if (tid == 0) n_valid = 0;
  __syncthreads();
  u32 mine = 0;
  for (u32 e = kHapK - 1 + tid; e < L; e += kHlqThreads) {  // k-mer ending at base e
    u32 code = 0;
    bool ok = true;
#pragma unroll
    for (int j = kHapK - 1; j >= 0; --j) {
      u32 const b = enc_base(s[e - j]);
      ok = ok && b < 4u;
      code = (code << 2) | (b & 3u);
    }
    if (ok) {
      atomicAdd(&cnt[padded(code)], 1u);
      mine++;
    }
  }
  mine = wave_sum_u32(mine);
  if (lane == 0 && mine) atomicAdd(&n_valid, mine);
  __syncthreads();
  u32 const valid = n_valid;
  if (valid == 0) {
    if (tid == 0) A.comp_hlq[ci] = 0.0;  // log1p(max(0, 0))
    return;
  }
  // ordered compaction of lgamma(count + 1) over the bins with count >= 2, forward and reverse-strand index order;
  // wave wv owns bins [4096 wv, 4096 wv + 4096)
  for (int strand = 0; strand < 2; ++strand) {
    u32 tot = 0;
    for (u32 it = 0; it < 64; ++it) {
      u32 const idx = wv * 4096u + it * 64u + lane;
      u32 const cc = cnt[padded(strand ? rc_code(idx, kHapK) : idx)];
      tot += static_cast<u32>(__popcll(__ballot(cc >= 2u)));
    }
    if (lane == 0) wave_tot[strand][wv] = tot;
  }
  __syncthreads();
  for (int strand = 0; strand < 2; ++strand) {
    u32 off = 0;
    for (u32 k = 0; k < wv; ++k) off += wave_tot[strand][k];
    for (u32 it = 0; it < 64; ++it) {
      u32 const idx = wv * 4096u + it * 64u + lane;
      u32 const cc = cnt[padded(strand ? rc_code(idx, kHapK) : idx)];
      u64 const m = __ballot(cc >= 2u);
      if (cc >= 2u) {
        u32 const at = off + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)));
        if (at < kListCap) list[strand][at] = A.t.lgam[cc];
      }
      off += static_cast<u32>(__popcll(m));
    }
  }
  __syncthreads();
  if (tid == 0) {
    f64 sc[2];
    for (int strand = 0; strand < 2; ++strand) {
      u32 const n = min(wave_tot[strand][0] + wave_tot[strand][1] + wave_tot[strand][2] + wave_tot[strand][3], kListCap);
      f64 sum = 0.0;
      for (u32 i = 0; i < n; ++i) sum += list[strand][i];
      sc[strand] = q_of(sum, valid, A.t.f7);
    }
    A.comp_hlq[ci] = log1p(fmax(0.0, fmax(sc[0], sc[1])));
  }
}

// ---- k_seqcx: one wavefront per variant ------------------------------------------------------------------
constexpr int kFlankK = 4;
constexpr u32 kFlankBins = 1u << (2 * kFlankK);

struct Flank {  // base/sequence_complexity.cpp:31-41 relative to the staged +-50 window
  i32 lo, hi;   // [lo, hi) inside the staged window (empty: lo == hi)
};
__device__ __forceinline__ Flank sub_flank(i32 L, i32 pos, i32 len, i32 flank, i32 base50) {
  i32 const s = max(0, pos - flank), e = min(L, pos + len + flank);
  Flank f;
  f.lo = s - base50;
  f.hi = s >= e ? f.lo : e - base50;
  return f;
}

// MaxHomopolymerRun (sequence_complexity.cpp:47-62) over win[f.lo, f.hi): every run start counts its run
__device__ i32 max_hrun(const u8* win, Flank f, u32 lane) {
  i32 best = 0;
  for (i32 i = f.lo + static_cast<i32>(lane); i < f.hi; i += 64) {
    if (i == f.lo || win[i] != win[i - 1]) {
      i32 j = i + 1;
      while (j < f.hi && win[j] == win[i]) ++j;
      best = max(best, j - i);
    }
  }
  return wave_max_i32(best);
}

// LocalShannonEntropy (sequence_complexity.cpp:75-120); the f32 operation order is the reference's
__device__ f32 shannon_entropy(const u8* win, Flank f, u32 lane) {
  u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (i32 i = f.lo + static_cast<i32>(lane); i < f.hi; i += 64) {
    u32 const b = enc_base(win[i]);
    c0 += b == 0u;
    c1 += b == 1u;
    c2 += b == 2u;
    c3 += b == 3u;
  }
  u32 const cnt[4] = {wave_sum_u32(c0), wave_sum_u32(c1), wave_sum_u32(c2), wave_sum_u32(c3)};
  if (f.hi <= f.lo) return 0.0f;
  f32 const total = static_cast<f32>(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
  if (total <= 0.0f) return 0.0f;
  f32 h = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (cnt[k] == 0) continue;
    f32 const freq = static_cast<f32>(cnt[k]) / total;
    h -= freq * log2f(freq);
  }
  return h;
}

// log1p(max(0, LongdustQ(k=4).Score(win[0, n)))) (longdust_scorer.h:246-330, sequence_complexity.cpp:396-397)
__device__ f64 flank_lq(const u8* win, i32 n, u32* cnt, const CxTables& t, u32 lane) {
  for (u32 i = lane; i < kFlankBins; i += 64) cnt[i] = 0;
  __builtin_amdgcn_wave_barrier();
  u32 mine = 0;
  for (i32 e = kFlankK - 1 + static_cast<i32>(lane); e < n; e += 64) {
    u32 code = 0;
    bool ok = true;
#pragma unroll
    for (int j = kFlankK - 1; j >= 0; --j) {
      u32 const b = enc_base(win[e - j]);
      ok = ok && b < 4u;
      code = (code << 2) | (b & 3u);
    }
    if (ok) {
      atomicAdd(&cnt[code], 1u);
      mine++;
    }
  }
  u32 const valid = wave_sum_u32(mine);
  __builtin_amdgcn_wave_barrier();
  if (valid == 0) return 0.0;  // log1p(0)
  f64 sc[2];
  for (int strand = 0; strand < 2; ++strand) {
    f64 sum = 0.0;
    for (u32 it = 0; it < kFlankBins / 64; ++it) {
      u32 const idx = it * 64u + lane;
      u32 const cc = cnt[strand ? rc_code(idx, kFlankK) : idx];
      f64 const v = cc >= 2u ? t.lgam[cc] : 0.0;
      u64 m = __ballot(cc >= 2u);
      while (m) {  // wave-uniform: every lane adds the same values in ascending index order
        u32 const l = static_cast<u32>(__builtin_ctzll(m));
        m &= m - 1;
        sum += read_lane_f64(v, l);
      }
    }
    sc[strand] = q_of(sum, valid, t.f4);
  }
  return log1p(fmax(0.0, fmax(sc[0], sc[1])));
}

// is win[st, st + p) a primitive motif? (sequence_complexity.cpp:130-148)
__device__ __forceinline__ bool primitive_motif(const u8* win, i32 st, i32 p) {
  for (i32 d = 1; d < p; ++d) {
    if (p % d != 0) continue;
    bool all = true;
    for (i32 i = d; i < p; ++i)
      if (win[st + i] != win[st + (i % d)]) {
        all = false;
        break;
      }
    if (all) return false;
  }
  return true;
}

struct TrBest {  // FlattenTRFeatures state of one (kind, period) lane (sequence_complexity.cpp:300-339)
  i32 dist, period, errors, span, stutter;
};
__device__ __forceinline__ void tr_fold(TrBest& b, i32 period, i32 start, i32 span, i32 errors, i32 vpos, i32 vlen) {
  i32 const tend = start + span, vend = vpos + vlen;
  i32 dist = 0;
  if (vpos >= start && vpos < tend) dist = 0;
  else if (vpos < start) dist = start - vend;
  else dist = vpos - tend;
  dist = max(0, dist);
  if (dist < b.dist) {
    b.dist = dist;
    b.period = period;
    b.errors = errors;
    b.span = span;
  }
  if (dist <= 1 && vlen > 0 && vlen <= period) b.stutter = 1;
}

// FindExactRepeats (:188-236) / FindApproxRepeats (:246-291) for ONE period, folded into FlattenTRFeatures.
// The reference walks the starts serially and jumps over every repeat it reports.  Here the 64 lanes evaluate 64
// candidate starts at once; the serial part -- which candidates the reference's loop actually visits -- is a
// wave-uniform walk over the ballot of hits: the lowest hit at or after `pos` is visited (every start before it
// just advances by one), is reported, and moves `pos` past the repeat.
__device__ void tr_period(const u8* win, i32 n, i32 p, bool approx, i32 vpos, i32 vlen, TrBest& b, u32 lane) {
  i32 pos = 0;
  for (i32 r0 = 0; r0 <= n - p; r0 += 64) {
    if (pos >= r0 + 64) continue;  // the whole round lies inside a repeat already reported
    i32 const start = r0 + static_cast<i32>(lane);
    bool hit = false;
    i32 span = 0, jump = 0, errors = 0;
    if (start <= n - p && start >= pos && (p == 1 || primitive_motif(win, start, p))) {
      if (!approx) {
        i32 match = p;
        while (start + match + p <= n) {
          bool same = true;
          for (i32 j = 0; j < p; ++j)
            if (win[start + match + j] != win[start + j]) {
              same = false;
              break;
            }
          if (!same) break;
          match += p;
        }
        i32 partial = 0;
        while (start + match + partial < n && partial < p && win[start + match + partial] == win[start + partial]) partial++;
        f32 const copies = static_cast<f32>(match + partial) / static_cast<f32>(p);
        hit = copies >= 2.5f;
        span = match + partial;
        jump = match;  // `start += match_len - 1` and the loop increment
      } else {
        span = p;
        while (start + span + p <= n) {
          i32 ue = 0;
          for (i32 j = 0; j < p; ++j) ue += win[start + span + j] != win[start + j];
          if (ue > 1) break;
          errors += ue;
          span += p;
        }
        f32 const copies = static_cast<f32>(span) / static_cast<f32>(p);
        f32 const purity = 1.0f - (static_cast<f32>(errors) / static_cast<f32>(span));
        hit = copies >= 3.0f && purity >= 0.75f;
        jump = span;
      }
    }
    u64 m = __ballot(hit);
    while (m) {
      u32 const l = static_cast<u32>(__builtin_ctzll(m));
      m &= m - 1;
      i32 const st = r0 + static_cast<i32>(l);
      if (st < pos) continue;  // inside the repeat reported just before
      tr_fold(b, p, st, __builtin_amdgcn_readlane(span, l), __builtin_amdgcn_readlane(errors, l), vpos, vlen);
      pos = st + __builtin_amdgcn_readlane(jump, l);
    }
  }
}

struct SeqCxAcc {  // SequenceComplexity (sequence_complexity.h:106-158)
  i32 ctx_hrun, delta_hrun, tr_period, stutter;
  f32 ctx_entropy, delta_entropy, tr_affinity, tr_purity;
  f64 ctx_flank_lq, ctx_hap_lq, delta_flank_lq;
};

constexpr int kCxWaves = 4;

__global__ __launch_bounds__(64 * kCxWaves) void k_seqcx(CxArgs A) {
  extern __shared__ unsigned char cx_lds[];
  u32 const lane = lane_id(), wv = threadIdx.x >> 6;
  int const w = blockIdx.x;
  u32 const nv = A.v.win_nvars[w];
  if (nv == 0) return;
  u8* win = cx_lds + static_cast<size_t>(wv) * A.ML;                                 // staged +-50 window
  u32* cnt = reinterpret_cast<u32*>(cx_lds + static_cast<size_t>(kCxWaves) * A.ML) + wv * kFlankBins;
  for (u32 i = wv; i < nv; i += kCxWaves) {
    size_t const vi = static_cast<size_t>(w) * A.MV + i;
    u32 const c = A.v.var_comp[vi];
    size_t const ci = static_cast<size_t>(w) * A.MC + c;
    u32 const hap0 = A.a.comp_hap0[ci], nh = A.a.comp_nhaps[ci];
    i32 const ref_pos = static_cast<i32>(A.v.var_ref_start[vi]), ref_len = static_cast<i32>(A.v.var_ref_len[vi]);
    u32 const nalts = A.v.var_nalts[vi];
    // ---- REF side of Score(): context + the REF halves of the deltas (same for every site of the variant) ----
    i32 r_hrun20, r_hrun5;
    f32 r_ent20, r_ent10;
    f64 r_lq;
    {
      size_t const hi = static_cast<size_t>(w) * A.MH + hap0;
      i32 const L = static_cast<i32>(A.a.hap_len[hi]);
      const u8* s = A.a.hap_bases + hi * A.ML;
      i32 const b50 = max(0, ref_pos - 50);
      Flank const f50 = sub_flank(L, ref_pos, ref_len, 50, b50);
      for (i32 k = static_cast<i32>(lane); k < f50.hi; k += 64) win[k] = s[b50 + k];
      __builtin_amdgcn_wave_barrier();
      r_hrun20 = max_hrun(win, sub_flank(L, ref_pos, ref_len, 20, b50), lane);
      r_ent20 = shannon_entropy(win, sub_flank(L, ref_pos, ref_len, 20, b50), lane);
      r_hrun5 = max_hrun(win, sub_flank(L, ref_pos, ref_len, 5, b50), lane);
      r_ent10 = shannon_entropy(win, sub_flank(L, ref_pos, ref_len, 10, b50), lane);
      r_lq = flank_lq(win, f50.hi, cnt, A.t, lane);
      __builtin_amdgcn_wave_barrier();
    }
    f64 const hap_lq = A.comp_hlq[ci];
    SeqCxAcc acc = {};  // var.mSeqCx starts all-zero; MergeMax only raises it (variant_annotator.cpp:68)
    bool any = false;
    // sites: (alt a, haplotype h >= 1 carrying it); if none, one pass with ALT == REF (variant_annotator.cpp:76-82)
    for (u32 site = 0; site <= nh; ++site) {
      u32 h;
      i32 apos, alen;
      if (site < nh) {
        if (site == 0) continue;
        u32 const al = A.v.var_hap_allele[vi * A.MH + site];
        if (al == 0 || al > nalts) continue;
        h = site;
        apos = static_cast<i32>(A.v.var_hap_start[vi * A.MH + site]);
        alen = max(ref_len, static_cast<i32>(A.v.alt_len[vi * A.MA + (al - 1)]));
        any = true;
      } else {
        if (any) break;
        h = 0;
        apos = ref_pos;
        alen = ref_len;
      }
      size_t const hi = static_cast<size_t>(w) * A.MH + hap0 + h;
      i32 const L = static_cast<i32>(A.a.hap_len[hi]);
      const u8* s = A.a.hap_bases + hi * A.ML;
      i32 const b50 = max(0, apos - 50);
      Flank const f50 = sub_flank(L, apos, alen, 50, b50);
      for (i32 k = static_cast<i32>(lane); k < f50.hi; k += 64) win[k] = s[b50 + k];
      __builtin_amdgcn_wave_barrier();
      SeqCxAcc cur;
      cur.ctx_hrun = r_hrun20;
      cur.ctx_entropy = r_ent20;
      cur.ctx_flank_lq = r_lq;
      cur.ctx_hap_lq = hap_lq;
      cur.delta_hrun = max_hrun(win, sub_flank(L, apos, alen, 5, b50), lane) - r_hrun5;
      cur.delta_entropy = shannon_entropy(win, sub_flank(L, apos, alen, 10, b50), lane) - r_ent10;
      cur.delta_flank_lq = flank_lq(win, f50.hi, cnt, A.t, lane) - r_lq;
      // ScoreTrMotif (sequence_complexity.cpp:442-461): results order = exact periods 1..6, then approximate 1..6;
      // the first strictly smaller distance wins, so folding in that order reproduces FlattenTRFeatures
      TrBest tb = {0x7fffffff, 0, 0, 0, 0};
      i32 const vpos = apos - b50;
      for (int kind = 0; kind < 2; ++kind)
        for (i32 p = 1; p <= 6 && p <= f50.hi; ++p) tr_period(win, f50.hi, p, kind == 1, vpos, alen, tb, lane);
      i32 const bd = tb.dist, bp = tb.period, be = tb.errors, bs = tb.span, st = tb.stutter;
      if (bd == 0x7fffffff) {
        cur.tr_affinity = 0.0f;
        cur.tr_purity = 0.0f;
        cur.tr_period = 0;
      } else {
        cur.tr_affinity = 1.0f / (1.0f + static_cast<f32>(bd));
        cur.tr_purity = bs <= 0 ? 0.0f : 1.0f - (static_cast<f32>(be) / static_cast<f32>(bs));
        cur.tr_period = bp;
      }
      cur.stutter = st;
      __builtin_amdgcn_wave_barrier();
      if (site < nh) {  // MergeMax (sequence_complexity.cpp:489-507)
        acc.ctx_hrun = max(acc.ctx_hrun, cur.ctx_hrun);
        acc.ctx_entropy = fmaxf(acc.ctx_entropy, cur.ctx_entropy);
        acc.ctx_flank_lq = fmax(acc.ctx_flank_lq, cur.ctx_flank_lq);
        acc.ctx_hap_lq = fmax(acc.ctx_hap_lq, cur.ctx_hap_lq);
        acc.delta_hrun = max(acc.delta_hrun, cur.delta_hrun);
        acc.delta_entropy = fmaxf(acc.delta_entropy, cur.delta_entropy);
        acc.delta_flank_lq = fmax(acc.delta_flank_lq, cur.delta_flank_lq);
        acc.tr_affinity = fmaxf(acc.tr_affinity, cur.tr_affinity);
        acc.tr_purity = fmaxf(acc.tr_purity, cur.tr_purity);
        acc.tr_period = max(acc.tr_period, cur.tr_period);
        acc.stutter = max(acc.stutter, cur.stutter);
      } else {
        acc = cur;
      }
    }
    if (lane == 0) {
      A.o.seq_cx_i[vi * 4 + 0] = acc.ctx_hrun;
      A.o.seq_cx_i[vi * 4 + 1] = acc.delta_hrun;
      A.o.seq_cx_i[vi * 4 + 2] = acc.tr_period;
      A.o.seq_cx_i[vi * 4 + 3] = acc.stutter;
      A.o.seq_cx_f[vi * 4 + 0] = acc.ctx_entropy;
      A.o.seq_cx_f[vi * 4 + 1] = acc.delta_entropy;
      A.o.seq_cx_f[vi * 4 + 2] = acc.tr_affinity;
      A.o.seq_cx_f[vi * 4 + 3] = acc.tr_purity;
      A.o.seq_cx_d[vi * 3 + 0] = acc.ctx_flank_lq;
      A.o.seq_cx_d[vi * 3 + 1] = acc.ctx_hap_lq;
      A.o.seq_cx_d[vi * 3 + 2] = acc.delta_flank_lq;
      // GraphComplexity::GraphEntanglementIndex (cbdg/graph_complexity.h:160-166), variant_annotator.cpp:87-99
      f64 const cc = static_cast<f64>(A.a.comp_cx[ci * 3 + 0]), bp2 = static_cast<f64>(A.a.comp_cx[ci * 3 + 1]);
      f64 const raw = (cc * bp2 * A.a.comp_cxf[ci * 4 + 1]) / (A.a.comp_cxf[ci * 4 + 0] + 1e-6);
      A.o.graph_cx[vi * 3 + 0] = log10(1.0 + raw);
      A.o.graph_cx[vi * 3 + 1] = A.a.comp_cxf[ci * 4 + 2];
      A.o.graph_cx[vi * 3 + 2] = static_cast<f64>(A.a.comp_cx[ci * 3 + 2]);
    }
  }
}

}  // namespace

int launch_annotate(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& v, double gc_frac,
                    const ma_cx_out_t& o) {
  ma_params_t const& p = ctx->prm;
  int const n = b.n_windows;
  if (n == 0) return MA_OK;
  if (!(gc_frac >= 0.0 && gc_frac <= 1.0)) gc_frac = std::clamp(gc_frac, 0.0, 1.0);  // longdust_scorer.h:228
  size_t const ML = static_cast<size_t>(p.max_hap_len);
  // tables: f4[ML+1] | f7[ML+1] | lgam[ML+2] | comp_hlq[n * MC]
  size_t const tab_f64 = (ML + 1) * 2 + (ML + 2);
  size_t const need = 8 * (tab_f64 + static_cast<size_t>(n) * p.max_comps) + 256;
  bool const rebuild = ctx->ws_cx.cap < need || ctx->cx_gc != gc_frac || ctx->cx_ml != p.max_hap_len;
  MA_HIP(ctx, ctx->ws_cx.reserve(need));
  f64* base = ctx->ws_cx.as<f64>();
  if (rebuild) {
    std::vector<f64>& h = ctx->cx_host;  // must outlive the async copy
    MA_HIP(ctx, ma_stream_sync(ctx));
    h.assign(tab_f64, 0.0);
    for (size_t l = 1; l <= ML; ++l) {
      h[l] = null_model_f(4, gc_frac, static_cast<int>(l));
      h[(ML + 1) + l] = null_model_f(7, gc_frac, static_cast<int>(l));
    }
    for (size_t c = 0; c < ML + 2; ++c) h[2 * (ML + 1) + c] = std::lgamma(static_cast<f64>(c + 1));
    MA_HIP(ctx, hipMemcpyAsync(base, h.data(), 8 * tab_f64, hipMemcpyHostToDevice, ctx->stream));
    ctx->cx_gc = gc_frac;
    ctx->cx_ml = p.max_hap_len;
  }
  CxArgs A;
  A.n_windows = n;
  A.MC = p.max_comps; A.MH = p.max_haps; A.ML = p.max_hap_len; A.MV = p.max_vars; A.MA = p.max_alts;
  A.a = a; A.v = v; A.o = o;
  A.t.f4 = base;
  A.t.f7 = base + (ML + 1);
  A.t.lgam = base + 2 * (ML + 1);
  A.comp_hlq = base + tab_f64;
  ctx->tic("k_hap_lq");
  hipLaunchKernelGGL(k_hap_lq, dim3(static_cast<u32>(n) * p.max_comps), dim3(kHlqThreads), 0, ctx->stream, A);
  ctx->toc();
  size_t const lds = static_cast<size_t>(kCxWaves) * ML + static_cast<size_t>(kCxWaves) * kFlankBins * 4;
  ctx->tic("k_seqcx");
  hipLaunchKernelGGL(k_seqcx, dim3(static_cast<u32>(n)), dim3(64 * kCxWaves), lds, ctx->stream, A);
  ctx->toc();
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

}  // namespace ma
