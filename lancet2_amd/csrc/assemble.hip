// Host orchestration of Graph::BuildComponentResults (cbdg/graph.cpp:78-256) over a batch of windows:
// the k-cascade loop runs on the host, each attempt launching the build passes (build.hip) and the
// cleaning/enumeration kernel (clean.hip) for the windows that are still unresolved at that k.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "graph_ws.h"

namespace ma {

int run_build_pass(ma_ctx* ctx, const DBatch& b, GraphWs& ws, u32* counters_dev, int tc_log2_alloc);
int run_count_inst(ma_ctx* ctx, const DBatch& b, const GraphWs& ws, int win0, int nwin, u32* maxima_dev);
int run_select_active(ma_ctx* ctx, const GraphWs& ws, int win0, int nwin, const u32* gate_approx, u32* win_k,
                      u32* active, u32* n_active_dev);
int run_clean_pass(ma_ctx* ctx, const DBatch& b, const GraphWs& ws, const ma_asm_out_t& out);

namespace {

__global__ void k_init_out(ma_asm_out_t o, u32* win_flags, int n) {
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o.win_status[i] = MA_W_NO_HAPLOTYPE;
  o.win_k[i] = 0;
  o.win_ncomp[i] = 0;
  win_flags[i] = 0;
}

// The classifier stages a tile of 64 reads (bases + qualities) in LDS: reads of up to kMaxAsmRead bases.  A window that
// holds a longer one is skipped and says so (MA_W_READ_OVERFLOW) -- it used to fail the whole batch.
constexpr u32 kMaxAsmRead = 1024;
__global__ __launch_bounds__(256) void k_flag_long_reads(DBatch b, ma_asm_out_t o, u32* win_flags) {
  // one wavefront per window, a lane per read (a lane per window walked its 600 read offsets one after the other: 0.8 ms)
  int const w = blockIdx.x * 4 + static_cast<int>(threadIdx.x >> 6);
  if (w >= b.n_windows) return;
  u32 ml = 0;
  for (u32 r = b.read_win_off[w] + (threadIdx.x & 63u); r < b.read_win_off[w + 1]; r += 64)
    ml = max(ml, static_cast<u32>(b.read_off[r + 1] - b.read_off[r]));
  if (__ballot(ml > kMaxAsmRead) != 0ull && (threadIdx.x & 63u) == 0) {
    o.win_status[w] = MA_W_NO_HAPLOTYPE | MA_W_READ_OVERFLOW;
    win_flags[w] = 1u;  // done: no pass plans for it or attempts it
  }
}

// Workload statistics for bench.py's algorithmic-byte model; launched only in ma_timing_control mode 3.
__global__ __launch_bounds__(256) void k_workload_stats(GraphWs ws, unsigned long long* acc) {
  int const a = blockIdx.x;
  const u64* keys = ws.tbl_key + (static_cast<size_t>(a) << ws.tc_log2);
  u32 const slots = 1u << ws.win_tc[a];
  u32 cnt = 0;
  for (u32 i = threadIdx.x; i < slots; i += 256) cnt += keys[i] != 0;
  __shared__ u32 sh[256];
  sh[threadIdx.x] = cnt;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < static_cast<u32>(d)) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], static_cast<unsigned long long>(sh[0]));              // distinct k-mers (N_raw)
    atomicAdd(&acc[1], static_cast<unsigned long long>(ws.n_nodes[a]));      // nodes that survive the first low-coverage pass
    atomicAdd(&acc[2], static_cast<unsigned long long>(ws.n_slow[a]));       // k-mer instances that took the hash-table path
    atomicAdd(&acc[3], static_cast<unsigned long long>(ws.win_ninst[ws.active[a]]));  // k-mer instances (N_inst)
  }
}

// Capacity retry: windows whose graph did not fit the node array / search arena are put back to "pending" so that the
// next pass, with larger capacities, assembles them from scratch; counts them.
__global__ void k_reset_overflowed(ma_asm_out_t o, u32* win_flags, int n, u32* count) {
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!(o.win_status[i] & MA_W_TABLE_OVERFLOW)) return;
  o.win_status[i] = MA_W_NO_HAPLOTYPE;
  o.win_k[i] = 0;
  o.win_ncomp[i] = 0;
  win_flags[i] = 0;
  atomicAdd(count, 1u);
}

int ceil_log2(u64 v) {
  int l = 0;
  while ((u64(1) << l) < v) ++l;
  return l;
}

u64 mod_inverse_pow2(u64 a) {  // a odd: Newton iteration mod 2^64
  u64 x = a;
  for (int i = 0; i < 6; ++i) x *= 2 - a * x;
  return x;
}

struct Carver {
  char* base;
  size_t off = 0;
  template <class T>
  T* take(size_t count) {
    off = (off + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

}  // namespace

int launch_assemble(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& out, const u32* gate_approx) {
  int const n = b.n_windows;
  if (n == 0) return MA_OK;
  ma_params_t const& P = ctx->prm;
  int const S = P.num_samples;

  // batch-level scratch: sequence bookkeeping + flags
  size_t const nseq = static_cast<size_t>(b.n_reads) + n;
  Carver cm{nullptr};
  auto carve_misc = [&](Carver& c, GraphWs& ws, u32** win_flags, u32** active, u32** counters) {
    ws.seq_inst_base = c.take<u32>(nseq + 1);
    ws.win_ninst = c.take<u32>(n);
    ws.win_nread_inst = c.take<u32>(n);
    *win_flags = c.take<u32>(n);
    *active = c.take<u32>(n);
    *counters = c.take<u32>(16);
    ws.rd_flag = c.take<u8>(static_cast<size_t>(b.n_reads) + 16);
  };
  GraphWs ws{};
  u32 *win_flags = nullptr, *active = nullptr, *counters = nullptr;
  carve_misc(cm, ws, &win_flags, &active, &counters);
  MA_HIP(ctx, ctx->ws_nodes.reserve(cm.off + 4096));
  Carver cm2{static_cast<char*>(ctx->ws_nodes.p)};
  carve_misc(cm2, ws, &win_flags, &active, &counters);
  ws.win_flags = win_flags;
  ws.num_samples = S;

  hipLaunchKernelGGL(k_init_out, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, out, win_flags, n);
  hipLaunchKernelGGL(k_flag_long_reads, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, b, out, win_flags);

  // capacity planning: instance maxima at the smallest k (the largest instance counts)
  ws.k = P.min_k;
  ws.win_k = out.win_k;  // (all zero here: every window is counted at the first rung)
  MA_TRY_RC(run_count_inst(ctx, b, ws, 0, n, counters));
  u32 maxima[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  MA_HIP(ctx, hipMemcpyAsync(maxima, counters, 32, hipMemcpyDeviceToHost, ctx->stream));
  MA_HIP(ctx, ma_stream_sync(ctx));
  u32 const max_inst = std::max<u32>(maxima[0], 1), max_read_inst = std::max<u32>(maxima[1], 1);
  u32 const max_refk = std::max<u32>(maxima[2], 1);
  ws.max_reads = std::max<u32>(maxima[3], 1);
  ws.max_read_len = std::max<u32>(maxima[4], 1);

  ws.tc_log2 = std::max(10, ceil_log2(static_cast<u64>(max_inst) * 4 / 3 + 16));
  ws.mc_log2 = std::max(10, ceil_log2(static_cast<u64>(max_read_inst) * 4 / 3 + 16));
  int const tc_log2_alloc = ws.tc_log2, mc_log2_alloc = ws.mc_log2;
  ws.inst_stride = (max_inst + 63) & ~63u;
  ws.ref_stride = (max_refk + 63) & ~63u;
  ws.max_ref_len = max_refk + static_cast<u32>(P.min_k) + 8;  // longest reference window (+ slack)
  u32 nc0 = std::max<u32>(8192, 2 * max_refk + 64);
  if (max_inst > 400000) nc0 = std::max<u32>(nc0, 32768);  // deep panels keep more recurrent-error k-mers
  if (const char* e = getenv("MA_NODE_CAP")) nc0 = std::max<u32>(256, static_cast<u32>(atoi(e)));  // tests: force the retry passes
  u32 ac0 = 1u << 16;
  if (const char* e = getenv("MA_ARENA_CAP")) ac0 = static_cast<u32>(atoi(e));

  // per-window workspace footprint -> chunk size
  auto carve_ws = [&](Carver& c, GraphWs& g, size_t A) {
    size_t const tcap = size_t(1) << g.tc_log2, mcap = size_t(1) << g.mc_log2, NC = g.nc;
    g.tbl_key = c.take<u64>(A * tcap);
    g.tbl_first = c.take<u32>(A * tcap);
    g.slot_node = c.take<u32>(A * tcap);
    g.tbl_cnt = c.take<u32>(A * tcap * (S + 2));
    g.inst_slot = c.take<u32>(A * g.inst_stride);
    g.slowq = c.take<u32>(A * g.inst_stride);
    g.n_slow = c.take<u32>(A);
    g.mm_mode = c.take<u32>(A);
    g.win_tc = c.take<u32>(A);
    g.mm_key = c.take<u64>(A * mcap);
    g.mm_min = c.take<u32>(A * mcap);
    g.n_nodes = c.take<u32>(A);
    g.nd_cnt = c.take<u32>(A * NC * S);
    g.nd_role = c.take<u32>(A * NC * 2);
    g.nd_src = c.take<u32>(A * NC);
    g.nd_label = c.take<u8>(A * NC);
    g.nd_sign = c.take<u8>(A * NC);
    g.nd_nedge = c.take<u8>(A * NC);
    g.nd_edge = c.take<u32>(A * NC * kEdgeCap);
    g.nd_ekey = c.take<u32>(A * NC * kEdgeCap);
    g.ref_node = c.take<u32>(A * g.ref_stride);
    g.ref_slot = c.take<u32>(A * g.ref_stride);
    g.nd_comp = c.take<u32>(A * NC);
    g.nd_len = c.take<u32>(A * NC);
    g.nd_alive = c.take<u8>(A * NC);
    g.nd_head = c.take<u32>(A * NC);
    g.nd_tail = c.take<u32>(A * NC);
    g.sl_next = c.take<u32>(A * NC);
    g.sl_prev = c.take<u32>(A * NC);
    g.sl_desc = c.take<u32>(A * NC);
    g.scratch = c.take<u32>(A * NC * 32);
    g.arena = c.take<uint4>(A * g.ac);
  };

  ctx->stats[2] += static_cast<unsigned long long>(n);
  // Pass 0 runs every window with the planned capacities.  Windows that come back flagged TABLE_OVERFLOW (graph larger
  // than the node array, walk search larger than the arena: the reference has no such limits) are re-assembled from
  // scratch by up to two more passes with 4x / 16x the node capacity and 8x / 64x the search arena -- only they are active, the workspace
  // is re-carved for fewer windows in flight.  What is still flagged afterwards is a limit growing cannot lift (more
  // than 16 edges on a node, a traversal cap the folded search cannot place: DESIGN.md section 7).
  for (int pass = 0; pass < 3; ++pass) {
    u32 const grow = pass == 0 ? 1u : (pass == 1 ? 4u : 16u);
    ws.nc = nc0 * grow;
    // the arena of the last pass holds what the reference's own cap allows: 2^20 pops (max_flow.h:69), a few pushes each
    // (components of more than ~400 nodes search unfolded, clean.hip)
    ws.ac = pass == 2 ? std::max<u32>(ac0 * 16u, 1u << 22) : ac0 * (pass == 1 ? 8u : 1u);
    ws.tc_log2 = tc_log2_alloc;
    ws.mc_log2 = mc_log2_alloc;
    if (pass > 0) {
      if (getenv("MA_NO_CAP_RETRY")) break;  // tests: show what a pass leaves flagged
      MA_HIP(ctx, hipMemsetAsync(counters + 10, 0, 4, ctx->stream));
      hipLaunchKernelGGL(k_reset_overflowed, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, out, win_flags, n, counters + 10);
      u32 n_over = 0;
      MA_HIP(ctx, hipMemcpyAsync(&n_over, counters + 10, 4, hipMemcpyDeviceToHost, ctx->stream));
      MA_HIP(ctx, ma_stream_sync(ctx));
      if (n_over == 0) break;
      if (getenv("MA_VERBOSE")) fprintf(stderr, "[microasm] assemble: capacity retry %d for %u windows (nc %u, ac %u)\n", pass, n_over, ws.nc, ws.ac);
    }
    Carver probe{nullptr};
    {
      GraphWs tmp = ws;
      carve_ws(probe, tmp, 1);
    }
    size_t const per_window = probe.off + 4096;
    size_t budget = stage_budget(0.30, ctx->ws_build.cap, size_t(24) << 30, ctx->hbm_share);
    if (const char* e = getenv("MA_WS_GB")) budget = static_cast<size_t>(atoi(e)) << 30;
    int chunk = static_cast<int>(std::max<size_t>(1, std::min<size_t>(n, budget / per_window)));
    MA_HIP(ctx, ctx->ws_build.reserve(per_window * static_cast<size_t>(chunk)));
    if (getenv("MA_VERBOSE"))
      fprintf(stderr, "[microasm] assemble: %d windows, %.2f MB/window, budget %.1f GB -> chunks of %d (nc %u, tc_log2 %d)\n", n,
              per_window / 1048576.0, budget / 1073741824.0, chunk, ws.nc, ws.tc_log2);

    for (int win0 = 0; win0 < n; win0 += chunk) {
      int const nwin = std::min(chunk, n - win0);
      ws.tc_log2 = tc_log2_alloc;
      ws.mc_log2 = mc_log2_alloc;
      Carver cw{static_cast<char*>(ctx->ws_build.p)};
      carve_ws(cw, ws, static_cast<size_t>(nwin));
      // one pass per rung a window still has to climb, not per rung of the ladder: every pending window attempts its own
      // next k (k_select_active); a second pass only sees the windows whose graph at that k had a cycle / was too complex
      int const rungs = (P.max_k - P.min_k) / P.k_step + 1;
      for (int kpass = 0; kpass <= rungs; ++kpass) {
        MA_TRY_RC(run_select_active(ctx, ws, win0, nwin, gate_approx, out.win_k, active, counters + 8));
        u32 host_cnt[2] = {0, 0};
        MA_HIP(ctx, hipMemcpyAsync(host_cnt, counters + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
        MA_HIP(ctx, ma_stream_sync(ctx));
        ws.n_active = static_cast<int>(host_cnt[0]);
        ws.active = active;
        if (host_cnt[1] == 0) break;  // every window of the chunk is resolved (graph.cpp:106 loop exit)
        if (ws.n_active > 0) {
          ctx->stats[3] += static_cast<unsigned long long>(ws.n_active);
          MA_TRY_RC(run_count_inst(ctx, b, ws, win0, nwin, counters));
          ws.tc_log2 = tc_log2_alloc;
          ws.mc_log2 = mc_log2_alloc;  // run_build_pass shrinks both to what this attempt needs
          MA_TRY_RC(run_build_pass(ctx, b, ws, counters + 12, tc_log2_alloc));
          if (ctx->collect) {
            MA_HIP(ctx, ctx->dev_stats.reserve(64));
            if (!ctx->dev_stats_clean) {
              MA_HIP(ctx, hipMemsetAsync(ctx->dev_stats.p, 0, 64, ctx->stream));
              ctx->dev_stats_clean = true;
            }
            hipLaunchKernelGGL(k_workload_stats, dim3(ws.n_active), dim3(256), 0, ctx->stream, ws,
                               ctx->dev_stats.as<unsigned long long>());
          }
          MA_TRY_RC(run_clean_pass(ctx, b, ws, out));
        }
      }
    }
  }
  return MA_OK;
}

}  // namespace ma
