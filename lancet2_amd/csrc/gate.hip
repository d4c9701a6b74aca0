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
// Mapping: one 256-thread workgroup per window; the window (<= 8 KB) is staged in LDS; lane t walks
// diagonals t+1, t+257, ...  (a wave's 64 lanes walk 64 adjacent diagonals, so the LDS reads are a
// broadcast of s[p] plus 64 consecutive bytes).  State per lane: the positions of the last 4
// mismatches.  HBM traffic: W bytes in, 8 bytes out per window.
#include "ma_internal.h"

namespace ma {

constexpr int kGateThreads = 256;
constexpr int kGateMaxW = 8192;

__global__ __launch_bounds__(kGateThreads) void gate_kernel(const u8* __restrict__ ref,
                                                            const u32* __restrict__ ref_off, int n_windows,
                                                            int mm, u32* __restrict__ out_approx,
                                                            u32* __restrict__ out_exact) {
  __shared__ u8 s[kGateMaxW + 64];
  __shared__ u32 red[2];
  int const w = blockIdx.x;
  if (w >= n_windows) return;
  u32 const beg = ref_off[w];
  int const W = min(static_cast<int>(ref_off[w + 1] - beg), kGateMaxW);
  for (int i = threadIdx.x; i < W; i += kGateThreads) s[i] = ref[beg + i];
  if (threadIdx.x < 2) red[threadIdx.x] = 0;
  __syncthreads();

  int best_m = 0, best_0 = 0;
  for (int d = 1 + static_cast<int>(threadIdx.x); d < W; d += kGateThreads) {
    int const len = W - d;
    // positions of the most recent mismatches on this diagonal: m0 newest ... m3 oldest
    int m0 = -1, m1 = -1, m2 = -1, m3 = -1;
    for (int p = 0; p < len; ++p) {
      bool const x = s[p] != s[p + d];
      if (x) {
        m3 = m2;
        m2 = m1;
        m1 = m0;
        m0 = p;
      }
      // longest window ending at p with <= mm mismatches starts after the (mm+1)-th newest one
      int const lim = mm == 0 ? m0 : (mm == 1 ? m1 : (mm == 2 ? m2 : m3));
      best_m = max(best_m, p - lim);
      best_0 = max(best_0, p - m0);
    }
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
  hipLaunchKernelGGL(gate_kernel, dim3(b.n_windows), dim3(kGateThreads), 0, ctx->stream, b.ref_bases,
                     b.ref_off, b.n_windows, ctx->prm.max_mismatch, max_approx, max_exact);
  ctx->toc();
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

}  // namespace ma
