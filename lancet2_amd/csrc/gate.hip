// Reference repeat gate on gfx950.
//
// Replaces base::HasRepeat / HasExactRepeat (base/repeat.cpp:348-375) as called by the k-cascade
// (cbdg/graph.h:127-131, budget 2) and the window skip (core/variant_builder.cpp:116-117).
// The reference evaluates O(n^2) k-mer pairs PER k.  All pairs (i, j) lie on diagonals d = j - i,
// and a k-window with <= m mismatches on a diagonal contains a (k-1)-window with <= m, so
//     HasRepeat(k, m)  <=>  max_d L_d(m) >= k,
// where L_d(m) is the longest run on diagonal d with <= m mismatches.  One pass over the W^2/2
// byte pairs therefore answers the gate for every k of the cascade at once.
//
// Mapping: one 256-thread workgroup per window; the window is staged in LDS FOUR times, copy c shifted by
// c bytes, so that the four bytes s[p + d .. p + d + 3] of any diagonal d are ONE aligned word of copy (d & 3); lane t
// walks diagonals t+1, t+257, ... four positions per iteration (one broadcast word of s[p ..], one word of the
// shifted copy, xor, four byte tests).  State per lane: the positions of the last MM + 1 mismatches.  HBM traffic:
// W bytes in, 8 bytes out per window.
#include "ma_internal.h"

namespace ma {

constexpr int kGateThreads = 256;
constexpr int kGateMaxW = 8192;
constexpr int kGateFastW = 2560;                 // windows up to here (the reference CLI's -w tops out at 2500) get the copies
constexpr int kGateCopy = kGateFastW + 64;       // bytes per shifted copy (a multiple of 4)
constexpr int kGateLds = 4 * kGateCopy;          // >= kGateMaxW + 64: a longer window fits once, unshifted

#ifndef MA_GATE_PACKED
#define MA_GATE_PACKED 1  // (0: one diagonal per lane everywhere, as until round 6 -- developer A/B builds)
#endif
template <int MM>
__global__ __launch_bounds__(kGateThreads) void gate_kernel(const u8* __restrict__ ref,
                                                            const u32* __restrict__ ref_off, int n_windows,
                                                            u32* __restrict__ out_approx,
                                                            u32* __restrict__ out_exact) {
  // 10 KB instead of four copies of the longest window the kernel accepts (33 KB): eight workgroups per CU, not four
  static_assert(kGateLds >= kGateMaxW + 64, "the unshifted window must fit");
  __shared__ u32 s4[kGateLds / 4];
  __shared__ u32 red[2];
  u8* const s = reinterpret_cast<u8*>(s4);
  int const w = blockIdx.x;
  if (w >= n_windows) return;
  u32 const beg = ref_off[w];
  int const W = min(static_cast<int>(ref_off[w + 1] - beg), kGateMaxW);
  bool const fast = W <= kGateFastW;
  // copy c, byte q = base q + c (the last len % 4 positions of a diagonal are walked byte by byte: no padding is read)
  for (int i = threadIdx.x; i < W; i += kGateThreads) {
    u8 const v = ref[beg + i];
    if (fast) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (i >= c) s[c * kGateCopy + i - c] = v;
    } else {
      s[i] = v;
    }
  }
  if (threadIdx.x < 2) red[threadIdx.x] = 0;
  __syncthreads();

  int best_m = 0, best_0 = 0;
  // Windows with the shifted copies (every window the reference's CLI can make): TWO diagonals per lane, their states --
  // the mismatch positions and the two maxima, all below 2^15 -- in the 16-bit halves of one register each (v_pk_sub_i16 /
  // v_pk_max_i16, one v_bfi per state word): 13 vector instructions per position PAIR instead of 2 x 9.5.  Diagonal d + 1 is
  // one position shorter than d: at that one position its half takes no part (`keep`).
  if (fast && MA_GATE_PACKED) {
    typedef short pk16 __attribute__((ext_vector_type(2)));
    auto as_pk = [](u32 v) { return __builtin_bit_cast(pk16, v); };
    auto as_u = [](pk16 v) { return __builtin_bit_cast(u32, v); };
    u32 bm = 0, b0 = 0;  // packed maxima
    for (int dA = 1 + 2 * static_cast<int>(threadIdx.x); dA < W; dA += 2 * kGateThreads) {
      int const dB = dA + 1;
      int const lenA = W - dA, lenB = lenA - 1;
      u32 m[MM + 1];
#pragma unroll
      for (int x = 0; x <= MM; ++x) m[x] = 0xFFFFFFFFu;  // (-1, -1)
      // M: 0xFFFF in the half of a diagonal that mismatches at p; keep: 0 in the half of a diagonal that has ended
      auto step2 = [&](int p, u32 M, u32 keep) {
        u32 const P = static_cast<u32>(p) * 0x10001u;
#pragma unroll
        for (int y = MM; y > 0; --y) m[y] = (M & m[y - 1]) | (~M & m[y]);
        m[0] = (M & P) | (~M & m[0]);
        bm = as_u(__builtin_elementwise_max(as_pk(bm), as_pk(as_u(as_pk(P) - as_pk(m[MM])) & keep)));
        b0 = as_u(__builtin_elementwise_max(as_pk(b0), as_pk(as_u(as_pk(P) - as_pk(m[0])) & keep)));
      };
      const u32* const shA = s4 + (dA & 3) * (kGateCopy / 4) + (dA >> 2);
      const u32* const shB = s4 + (dB & 3) * (kGateCopy / 4) + (dB >> 2);
      int p = 0;
      for (; p + 4 <= lenB; p += 4) {
        u32 const a = s4[p >> 2];
        u32 const xA = a ^ shA[p >> 2], xB = a ^ shB[p >> 2];
        // bit 7 of every byte that is not zero (the other bits do not matter) ...
        u32 const fA = ((xA & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xA;
        u32 const fB = ((xB & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xB;
        // ... spread over a half each by v_perm_b32's sign selectors (8 / 9: bit 15 / 31 of the second operand, 10 / 11: of
        // the first): one instruction per position for both diagonals' masks
        u32 const gA = fA << 8, gB = fB << 8;
        step2(p, __builtin_amdgcn_perm(gB, gA, 0x0A0A0808u), 0xFFFFFFFFu);      // byte 0: bit 7 -> 15
        step2(p + 1, __builtin_amdgcn_perm(fB, fA, 0x0A0A0808u), 0xFFFFFFFFu);  // byte 1: bit 15
        step2(p + 2, __builtin_amdgcn_perm(gB, gA, 0x0B0B0909u), 0xFFFFFFFFu);  // byte 2: bit 23 -> 31
        step2(p + 3, __builtin_amdgcn_perm(fB, fA, 0x0B0B0909u), 0xFFFFFFFFu);  // byte 3: bit 31
      }
      for (; p < lenA; ++p) {
        u32 const nA = s[p] != s[p + dA] ? 0xFFFFu : 0u;
        bool const liveB = p < lenB;
        u32 const nB = (liveB && s[p] != s[p + dB]) ? 0xFFFF0000u : 0u;
        step2(p, nA | nB, liveB ? 0xFFFFFFFFu : 0x0000FFFFu);
      }
    }
    best_m = max(static_cast<int>(bm & 0xFFFFu), static_cast<int>(bm >> 16));
    best_0 = max(static_cast<int>(b0 & 0xFFFFu), static_cast<int>(b0 >> 16));
  } else
  for (int d = 1 + static_cast<int>(threadIdx.x); d < W; d += kGateThreads) {
    int const len = W - d;
    // positions of the most recent mismatches on this diagonal: m[0] newest ... m[MM] oldest kept
    int m[MM + 1];
#pragma unroll
    for (int x = 0; x <= MM; ++x) m[x] = -1;
    auto step = [&](int p, bool x) {
      if (x) {
#pragma unroll
        for (int y = MM; y > 0; --y) m[y] = m[y - 1];
        m[0] = p;
      }
      // longest window ending at p with <= MM mismatches starts after the (MM+1)-th newest one
      best_m = max(best_m, p - m[MM]);
      best_0 = max(best_0, p - m[0]);
    };
    int p = 0;
    if (fast) {
      const u32* const shifted = s4 + (d & 3) * (kGateCopy / 4) + (d >> 2);  // word q of it = s[4 q + d ..]
      for (; p + 4 <= len; p += 4) {
        u32 const x = s4[p >> 2] ^ shifted[p >> 2];
        step(p, (x & 0xFFu) != 0);
        step(p + 1, (x & 0xFF00u) != 0);
        step(p + 2, (x & 0xFF0000u) != 0);
        step(p + 3, (x & 0xFF000000u) != 0);
      }
    }
    for (; p < len; ++p) step(p, s[p] != s[p + d]);
  }
  atomicMax(&red[0], static_cast<u32>(best_m));
  atomicMax(&red[1], static_cast<u32>(best_0));
  __syncthreads();
  if (threadIdx.x == 0) {
    out_approx[w] = red[0];
    out_exact[w] = red[1];
  }
}

int launch_gate(ma_ctx* ctx, const DBatch& b, u32* max_approx, u32* max_exact) {
  if (b.n_windows == 0) return MA_OK;
  ctx->tic("gate_kernel");
  int const mm = ctx->prm.max_mismatch;  // (0..3: checked by ma_create)
  auto kern = mm == 0 ? gate_kernel<0> : (mm == 1 ? gate_kernel<1> : (mm == 2 ? gate_kernel<2> : gate_kernel<3>));
  hipLaunchKernelGGL(kern, dim3(b.n_windows), dim3(kGateThreads), 0, ctx->stream, b.ref_bases, b.ref_off, b.n_windows,
                     max_approx, max_exact);
  ctx->toc();
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

}  // namespace ma
