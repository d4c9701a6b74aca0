// Host orchestration of Graph::BuildComponentResults (cbdg/graph.cpp:78-256) over a batch of windows:
// the k-cascade loop runs on the host, each attempt launching the build passes (build.hip) and the
// cleaning/enumeration kernel (clean.hip) for the windows that are still unresolved at that k.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "graph_ws.h"

namespace ma {

int run_build_pass(ma_ctx* ctx, const DBatch& b, GraphWs& ws, u32* counters_dev, int tc_log2_alloc);
int run_count_inst(ma_ctx* ctx, const DBatch& b, const GraphWs& ws, int win0, int nwin, u32* maxima_dev);
int run_select_active(ma_ctx* ctx, const GraphWs& ws, int win0, int nwin, const u32* gate_approx, u32* win_k,
                      u32* active, u32* n_active_dev);
int run_clean_pass(ma_ctx* ctx, const DBatch& b, const GraphWs& ws, const ma_asm_out_t& out);

namespace {

__global__ void k_init_out(ma_asm_out_t o, u32* win_flags, int n, u32 MC, u32 MH) {
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o.win_status[i] = MA_W_NO_HAPLOTYPE;
  o.win_k[i] = 0;
  o.win_ncomp[i] = 0;
  win_flags[i] = 0;
  // the per-component / per-haplotype records start from zero: slots a window does not use read the same whatever the
  // buffers held before (the host route copies these small arrays whole)
  size_t const c0 = static_cast<size_t>(i) * MC, h0 = static_cast<size_t>(i) * MH;
  for (u32 c = 0; c < MC; ++c) {
    if (o.comp_anchor) o.comp_anchor[c0 + c] = 0;
    if (o.comp_hap0) o.comp_hap0[c0 + c] = 0;
    if (o.comp_nhaps) o.comp_nhaps[c0 + c] = 0;
    if (o.comp_cx)
      for (u32 x = 0; x < 3; ++x) o.comp_cx[(c0 + c) * 3 + x] = 0;
    if (o.comp_cxf)
      for (u32 x = 0; x < 4; ++x) o.comp_cxf[(c0 + c) * 4 + x] = 0.0;
  }
  for (u32 h = 0; h < MH; ++h) {
    if (o.hap_len) o.hap_len[h0 + h] = 0;
    if (o.hap_nruns) o.hap_nruns[h0 + h] = 0;
    if (o.hap_stats)
      for (u32 x = 0; x < 6; ++x) o.hap_stats[(h0 + h) * 6 + x] = 0.0;
  }
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
  const u64* keys = ws.tbl_key + (static_cast<size_t>(a) << tbl_log2(ws));
  u32 const slots = ws.win_nslots[a];
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
    atomicAdd(&acc[4], static_cast<unsigned long long>(ws.n_edgeq[a]));      // the reads' (k+1)-mers queued for k_graph
    atomicAdd(&acc[5], static_cast<unsigned long long>(ws.n_genq[a]));       // read-support counts / keys queued (k_support)
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

// ---- speculative tail of the k ladder ------------------------------------------------------------------------------------
// A window whose graph has a cycle (or is too complex, or yields nothing) at k goes on to the next rung; a tandem duplication
// of 80 bases sends it up a dozen rungs.  One pass of the build + clean kernels per rung costs its fixed latency (~3.5 ms: the
// slowest window of k_clean, the kernels' minimum durations) however few windows it holds, and the tail of the ladder is a
// dozen such passes over a few dozen windows.  Once few windows are pending, their NEXT SIX RUNGS are therefore attempted at
// once: the pending windows are copied into a derived batch, one copy per rung, the derived batch is assembled by this same
// function (every copy at its one k, k_select_active: win_kfirst), and each window takes the result of the first rung that
// resolved -- what the rung-by-rung loop would have stopped at.  Rungs beyond it were wasted work, on idle hardware.
struct SpecDesc {  // one pending window, as the host needs it to lay the derived batch out
  u32 w, k, gate, ref_len, n_reads, pad_;
  u64 read_bytes;
};
struct SpecVirt {  // one window of the derived batch
  u32 src_w;       // the window it is a copy of
  u32 k;           // its rung
  u32 ref_off;     // where its reference bytes start in the derived batch
  u32 rwo;         // its first read's index there
  u64 byte_off;    // its first read byte there
};
struct SpecBatch {  // the derived batch's arrays (device, writable)
  u8* ref_bases;
  u32* ref_off;
  u32* read_win_off;
  u64* read_off;
  u8* read_bases;
  u8* read_quals;
  u32* read_qname_id;
  u8* read_sample;
  u8* read_flags;
  i32* read_hint;
};
__global__ void k_spec_describe(DBatch b, const u32* active, u32 n, const u32* win_k, const u32* gate, SpecDesc* out) {
  u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 const w = active[i];
  u32 const r0 = b.read_win_off[w], r1 = b.read_win_off[w + 1];
  out[i] = SpecDesc{w, win_k[w], gate[w], b.ref_off[w + 1] - b.ref_off[w], r1 - r0, 0u, b.read_off[r1] - b.read_off[r0]};
}
__global__ __launch_bounds__(256) void k_spec_gather(DBatch b, const SpecVirt* virt, u32 V, SpecBatch d) {
  u32 const v = blockIdx.x;
  SpecVirt const sv = virt[v], nx = virt[v + 1];  // (virt[V] holds the totals)
  u32 const w = sv.src_w;
  u32 const ref_len = nx.ref_off - sv.ref_off, nr = nx.rwo - sv.rwo;
  u64 const nbytes = nx.byte_off - sv.byte_off;
  const u8* rs = b.ref_bases + b.ref_off[w];
  for (u32 i = threadIdx.x; i < ref_len; i += 256) d.ref_bases[sv.ref_off + i] = rs[i];
  u32 const r0 = b.read_win_off[w];
  u64 const byte0 = b.read_off[r0];
  for (u32 i = threadIdx.x; i < nr; i += 256) {
    u32 const r = r0 + i, at = sv.rwo + i;
    d.read_off[at] = sv.byte_off + (b.read_off[r] - byte0);
    d.read_qname_id[at] = b.read_qname_id[r];
    d.read_sample[at] = b.read_sample[r];
    d.read_flags[at] = b.read_flags[r];
    if (d.read_hint) d.read_hint[at] = b.read_hint[r];
  }
  for (u64 i = threadIdx.x; i < nbytes; i += 256) {
    d.read_bases[sv.byte_off + i] = b.read_bases[byte0 + i];
    d.read_quals[sv.byte_off + i] = b.read_quals[byte0 + i];
  }
  if (threadIdx.x == 0) {
    d.ref_off[v] = sv.ref_off;
    d.read_win_off[v] = sv.rwo;
    if (v + 1 == V) {
      d.ref_off[V] = nx.ref_off;
      d.read_win_off[V] = nx.rwo;
      d.read_off[nx.rwo] = nx.byte_off;
    }
  }
}
struct SpecField {  // one output array: bytes per window, in the derived batch's outputs and in the caller's
  const u8* src;
  u8* dst;
  u32 bytes;
};
struct SpecFields {
  SpecField f[13];
  u32 n;
};
// a workgroup per pending window: the first of its copies (rung order) that resolved hands its outputs over
__global__ __launch_bounds__(256) void k_spec_select(const SpecDesc* desc, const u32* first_virt, const SpecVirt* virt, u32 P,
                                                     ma_asm_out_t inner, SpecFields F, ma_asm_out_t outer, u32* win_flags, u32* counts) {
  u32 const p = blockIdx.x;
  if (p >= P) return;
  u32 const w = desc[p].w;
  u32 const v0 = first_virt[p], v1 = first_virt[p + 1];
  u32 chosen = 0xFFFFFFFFu;
  for (u32 v = v0; v < v1 && chosen == 0xFFFFFFFFu; ++v)
    if (inner.win_ncomp[v] > 0 || (inner.win_status[v] & MA_W_TABLE_OVERFLOW)) chosen = v;
  if (chosen == 0xFFFFFFFFu) {  // none of these rungs: still pending, next time from the last one tried
    if (threadIdx.x == 0) {
      outer.win_k[w] = virt[v1 - 1].k;
      atomicAdd(&counts[0], v1 - v0);   // rungs the rung-by-rung loop would have attempted too
      atomicAdd(&counts[1], 1u);        // still pending
    }
    return;
  }
  for (u32 x = 0; x < F.n; ++x) {
    const u8* s = F.f[x].src + static_cast<size_t>(chosen) * F.f[x].bytes;
    u8* d = F.f[x].dst + static_cast<size_t>(w) * F.f[x].bytes;
    if ((F.f[x].bytes & 3u) == 0)
      for (u32 i = threadIdx.x; i < F.f[x].bytes / 4u; i += 256) reinterpret_cast<u32*>(d)[i] = reinterpret_cast<const u32*>(s)[i];
    else
      for (u32 i = threadIdx.x; i < F.f[x].bytes; i += 256) d[i] = s[i];
  }
  if (threadIdx.x == 0) {
    atomicOr(&win_flags[w], 1u);
    atomicAdd(&counts[0], chosen - v0 + 1u);
  }
}

constexpr u32 kSpecPendingCap = 2048;
static u32 const kSpecPending = getenv("MA_SPEC_PENDING") ? static_cast<u32>(atoi(getenv("MA_SPEC_PENDING"))) : 256u;  // windows pending at most
static u32 const kSpecRungs = getenv("MA_SPEC_RUNGS") ? static_cast<u32>(atoi(getenv("MA_SPEC_RUNGS"))) : 6u;          // rungs per round

// `active[0, P)` = the pending windows of the chunk, win_k[w] = the rung each of them would attempt next (k_select_active
// has just run).  Resolves them as far as kSpecRungs rungs go; *still = how many remain pending.
int speculate_tail(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& out, const u32* gate_approx, const u32* active, u32 P,
                   u32* win_flags, u32* counters, u32* still) {
  ma_params_t const& prm = ctx->prm;
  // 1. what the pending windows look like
  MA_HIP(ctx, ctx->spec_data.reserve(sizeof(SpecDesc) * kSpecPendingCap + 4096));
  SpecDesc* d_desc = ctx->spec_data.as<SpecDesc>();
  hipLaunchKernelGGL(k_spec_describe, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, b, active, P, out.win_k, gate_approx, d_desc);
  std::vector<SpecDesc> desc(P);
  MA_HIP(ctx, hipMemcpyAsync(desc.data(), d_desc, sizeof(SpecDesc) * P, hipMemcpyDeviceToHost, ctx->stream));
  MA_HIP(ctx, ma_stream_sync(ctx));
  // 2. the derived batch: every pending window once per rung (the ladder's skip rule for gated k: k_select_active)
  std::vector<SpecVirt> virt;
  std::vector<u32> first_virt(P + 1, 0), vk;
  u32 ref_at = 0, read_at = 0;
  u64 byte_at = 0;
  for (u32 p = 0; p < P; ++p) {
    first_virt[p] = static_cast<u32>(virt.size());
    u32 k = desc[p].k;
    for (u32 r = 0; r < kSpecRungs && k <= static_cast<u32>(prm.max_k); ++r) {
      virt.push_back(SpecVirt{desc[p].w, k, ref_at, read_at, byte_at});
      vk.push_back(k);
      ref_at += desc[p].ref_len;
      read_at += desc[p].n_reads;
      byte_at += desc[p].read_bytes;
      k += static_cast<u32>(prm.k_step);
      if (k <= desc[p].gate) k += ((desc[p].gate - k) / static_cast<u32>(prm.k_step) + 1u) * static_cast<u32>(prm.k_step);
    }
  }
  u32 const V = static_cast<u32>(virt.size());
  first_virt[P] = V;
  virt.push_back(SpecVirt{0, 0, ref_at, read_at, byte_at});  // totals
  // device layout of the derived batch + its bookkeeping arrays
  Carver probe{nullptr};
  auto carve = [&](Carver& c, SpecBatch* d, SpecDesc** dd, SpecVirt** dv, u32** dfirst, u32** dk, u32** dgate) {
    *dd = c.take<SpecDesc>(kSpecPendingCap);
    *dv = c.take<SpecVirt>(V + 1);
    *dfirst = c.take<u32>(P + 1);
    *dk = c.take<u32>(V);
    *dgate = c.take<u32>(V);
    d->ref_bases = c.take<u8>(static_cast<size_t>(ref_at) + 128);
    d->ref_off = c.take<u32>(V + 1);
    d->read_win_off = c.take<u32>(V + 1);
    d->read_off = c.take<u64>(static_cast<size_t>(read_at) + 1);
    d->read_bases = c.take<u8>(byte_at + 128);
    d->read_quals = c.take<u8>(byte_at + 128);
    d->read_qname_id = c.take<u32>(static_cast<size_t>(read_at) + 1);
    d->read_sample = c.take<u8>(static_cast<size_t>(read_at) + 16);
    d->read_flags = c.take<u8>(static_cast<size_t>(read_at) + 16);
    d->read_hint = b.read_hint ? c.take<i32>(static_cast<size_t>(read_at) + 1) : nullptr;
  };
  SpecBatch sb{};
  SpecDesc* dd = nullptr;
  SpecVirt* dv = nullptr;
  u32 *dfirst = nullptr, *dk = nullptr, *dgate = nullptr;
  carve(probe, &sb, &dd, &dv, &dfirst, &dk, &dgate);
  MA_HIP(ctx, ctx->spec_data.reserve(probe.off + 4096));  // (may move the buffer: d_desc is not used again)
  Carver cv{static_cast<char*>(ctx->spec_data.p)};
  carve(cv, &sb, &dd, &dv, &dfirst, &dk, &dgate);
  MA_HIP(ctx, hipMemcpyAsync(dd, desc.data(), sizeof(SpecDesc) * P, hipMemcpyHostToDevice, ctx->stream));
  MA_HIP(ctx, hipMemcpyAsync(dv, virt.data(), sizeof(SpecVirt) * (V + 1), hipMemcpyHostToDevice, ctx->stream));
  MA_HIP(ctx, hipMemcpyAsync(dfirst, first_virt.data(), 4ull * (P + 1), hipMemcpyHostToDevice, ctx->stream));
  MA_HIP(ctx, hipMemcpyAsync(dk, vk.data(), 4ull * V, hipMemcpyHostToDevice, ctx->stream));
  MA_HIP(ctx, hipMemsetAsync(dgate, 0, 4ull * V, ctx->stream));
  MA_HIP(ctx, hipMemsetAsync(sb.ref_bases + ref_at, 0, 128, ctx->stream));
  MA_HIP(ctx, hipMemsetAsync(sb.read_bases + byte_at, 0, 128, ctx->stream));
  MA_HIP(ctx, hipMemsetAsync(sb.read_quals + byte_at, 0, 128, ctx->stream));
  hipLaunchKernelGGL(k_spec_gather, dim3(V), dim3(256), 0, ctx->stream, b, dv, V, sb);
  MA_HIP(ctx, ma_stream_sync(ctx));  // (the host vectors above are the sources of the copies)
  DBatch d2{};
  d2.n_windows = static_cast<int>(V);
  d2.n_reads = read_at;
  d2.ref_bases = sb.ref_bases; d2.ref_off = sb.ref_off; d2.read_win_off = sb.read_win_off; d2.read_off = sb.read_off;
  d2.read_bases = sb.read_bases; d2.read_quals = sb.read_quals; d2.read_qname_id = sb.read_qname_id;
  d2.read_sample = sb.read_sample; d2.read_flags = sb.read_flags; d2.read_hint = sb.read_hint;
  // 3. its outputs
  size_t const MC = prm.max_comps, MH = prm.max_haps, ML = prm.max_hap_len, MR = prm.max_runs;
  ma_asm_out_t o2{};
  SpecFields F{};
  {
    Carver oprobe{nullptr};
    auto carve_out = [&](Carver& c, ma_asm_out_t* o) {
      o->win_status = c.take<u32>(V); o->win_k = c.take<u32>(V); o->win_ncomp = c.take<u32>(V);
      o->comp_anchor = c.take<u32>(V * MC); o->comp_hap0 = c.take<u32>(V * MC); o->comp_nhaps = c.take<u32>(V * MC);
      o->comp_cx = c.take<u32>(V * MC * 3); o->comp_cxf = c.take<double>(V * MC * 4);
      o->hap_len = c.take<u32>(V * MH); o->hap_nruns = c.take<u32>(V * MH); o->hap_stats = c.take<double>(V * MH * 6);
      o->hap_bases = c.take<u8>(V * MH * ML); o->hap_runs = c.take<u32>(V * MH * MR * 2);
    };
    carve_out(oprobe, &o2);
    MA_HIP(ctx, ctx->spec_out.reserve(oprobe.off + 4096));
    Carver co{static_cast<char*>(ctx->spec_out.p)};
    carve_out(co, &o2);
    auto add = [&](const void* src, void* dst, size_t bytes) {
      if (src && dst) F.f[F.n++] = SpecField{static_cast<const u8*>(src), static_cast<u8*>(dst), static_cast<u32>(bytes)};
    };
    add(o2.win_status, out.win_status, 4); add(o2.win_k, out.win_k, 4); add(o2.win_ncomp, out.win_ncomp, 4);
    add(o2.comp_anchor, out.comp_anchor, 4 * MC); add(o2.comp_hap0, out.comp_hap0, 4 * MC); add(o2.comp_nhaps, out.comp_nhaps, 4 * MC);
    add(o2.comp_cx, out.comp_cx, 12 * MC); add(o2.comp_cxf, out.comp_cxf, 32 * MC);
    add(o2.hap_len, out.hap_len, 4 * MH); add(o2.hap_nruns, out.hap_nruns, 4 * MH); add(o2.hap_stats, out.hap_stats, 48 * MH);
    add(o2.hap_bases, out.hap_bases, MH * ML); add(o2.hap_runs, out.hap_runs, 8 * MH * MR);
  }
  // 4. the derived batch through this same stage, every window at its one rung.  The batch-level bookkeeping (flags, active
  //    list, counters: ws_nodes) gets a buffer of its own; the per-window workspace (ws_build) is the caller's chunk's --
  //    nothing of it is live between passes, and once the tail is attempted this way the chunk sees no ordinary pass again
  //    (a second workspace of that size, 20 GB per lane, left the stages that follow nothing to reserve).
  std::swap(ctx->ws_nodes, ctx->spec_nodes);
  ctx->spec_k = dk;
  int const rc = launch_assemble(ctx, d2, o2, dgate);
  ctx->spec_k = nullptr;
  std::swap(ctx->ws_nodes, ctx->spec_nodes);
  MA_TRY_RC(rc);
  // 5. every pending window takes the first rung that resolved
  MA_HIP(ctx, hipMemsetAsync(counters, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_spec_select, dim3(P), dim3(256), 0, ctx->stream, dd, dfirst, dv, P, o2, F, out, win_flags, counters);
  u32 host[2] = {0, 0};
  MA_HIP(ctx, hipMemcpyAsync(host, counters, 8, hipMemcpyDeviceToHost, ctx->stream));
  MA_HIP(ctx, ma_stream_sync(ctx));
  ctx->stats[3] += host[0];
  *still = host[1];
  if (getenv("MA_VERBOSE"))
    fprintf(stderr, "[microasm] assemble: ladder tail -- %u pending windows x up to %u rungs = %u attempts at once, %u still pending\n", P,
            kSpecRungs, V, host[1]);
  return MA_OK;
}

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
  bool const nested = ctx->spec_k != nullptr;  // the nested pass of speculate_tail: every window at its one rung
  ws.win_kfirst = ctx->spec_k;
  u32 *win_flags = nullptr, *active = nullptr, *counters = nullptr;
  carve_misc(cm, ws, &win_flags, &active, &counters);
  MA_HIP(ctx, ctx->ws_nodes.reserve(cm.off + 4096));
  Carver cm2{static_cast<char*>(ctx->ws_nodes.p)};
  carve_misc(cm2, ws, &win_flags, &active, &counters);
  ws.win_flags = win_flags;
  ws.num_samples = S;

  hipLaunchKernelGGL(k_init_out, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, out, win_flags, n, static_cast<u32>(P.max_comps),
                     static_cast<u32>(P.max_haps));
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
  // The k-mer table of a window is planned for its INSTANCES (every one could be a k-mer of its own).  A deep window (0.9 M
  // instances in a 2000x panel) holds some 40 k distinct k-mers: planned by instances its table is 2^21 slots = 67 MB, a lane's
  // share of the build budget holds 200 such windows and the lane pays the clean stage's latency once per chunk.  The first
  // pass therefore plans a quarter of the instances; a window whose table does fill up is flagged like any other capacity and
  // re-assembled by the retry passes, which plan the full table.
  int const tc_full = ws.tc_log2, mc_log2_alloc = ws.mc_log2;
  int tc_first = max_inst > 400000 ? std::max(17, ceil_log2(static_cast<u64>(max_inst) / 4 + 16)) : tc_full;
  if (const char* e = getenv("MA_TC_FIRST")) tc_first = std::max(10, atoi(e));  // tests: force the table retry
  tc_first = std::min(tc_first, tc_full);
  ws.inst_stride = (max_inst + 63) & ~63u;
  ws.ref_stride = (max_refk + 63) & ~63u;
  ws.max_ref_len = max_refk + static_cast<u32>(P.min_k) + 8;  // longest reference window (+ slack)
  u32 nc0 = std::max<u32>(8192, 2 * max_refk + 64);
  if (max_inst > 400000) nc0 = std::max<u32>(nc0, 32768);  // deep panels keep more recurrent-error k-mers
  if (const char* e = getenv("MA_NODE_CAP")) nc0 = std::max<u32>(256, static_cast<u32>(atoi(e)));  // tests: force the retry passes
  u32 ac0 = 1u << 16;
  if (const char* e = getenv("MA_ARENA_CAP")) ac0 = static_cast<u32>(atoi(e));
  ws.vc = 2048;
  ws.cg_sc = 2048;
  ws.pool_cap = 49152;

  // per-window workspace footprint -> chunk size
  size_t mm_share = 0;  // bytes of the mate-mer pool per window of the chunk (set per pass below)
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
    g.win_nslots = c.take<u32>(A);
    g.slow_rec = c.take<uint4>(A * g.inst_stride);
    g.gr_done = c.take<u32>(A);
    g.n_edgeq = c.take<u32>(A);
    g.n_genq = c.take<u32>(A);
    // HBM-resident mate-mer sets: a pool the windows that need one carve theirs out of (k_support).  A full set per window
    // when every window will need one (no mapping hints; capacity retries route every mate-mer through it), a token share
    // otherwise -- the pool is part of the chunk's budgeted workspace either way (round 4 reserved it on demand, outside the
    // budget: tens of GB for a chunk of deep windows without hints)
    g.mm_off = c.take<unsigned long long>(A);
    g.mm_log2 = c.take<u32>(A);
    g.mm_pool_used = c.take<unsigned long long>(1);
    g.mm_pool_bytes = static_cast<unsigned long long>(A) * mm_share;
    g.mm_pool = c.take<u8>(g.mm_pool_bytes + 256);
    (void)mcap;
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
    // compact graph (k_clean_chains -> k_clean_tail)
    size_t const VC = g.vc;
    g.cg_state = c.take<u32>(A);
    g.cg_hdr = c.take<u32>(A * kCgHdr);
    g.cg_cnt = c.take<u32>(A * VC * S);
    g.cg_role = c.take<u32>(A * VC * 2);
    g.cg_bsrc = c.take<u32>(A * VC);
    g.cg_blen = c.take<u32>(A * VC);
    g.cg_len = c.take<u32>(A * VC);
    g.cg_comp = c.take<u32>(A * VC);
    g.cg_label = c.take<u8>(A * VC);
    g.cg_sign = c.take<u8>(A * VC);
    g.cg_bsign = c.take<u8>(A * VC);
    g.cg_nedge = c.take<u8>(A * VC);
    g.cg_alive = c.take<u8>(A * VC);
    g.cg_edge = c.take<u32>(A * VC * kCgEdgeCap);
    g.cg_head = c.take<u32>(A * VC);
    g.cg_tail = c.take<u32>(A * VC);
    g.cg_snext = c.take<u32>(A * VC);
    g.cg_sprev = c.take<u32>(A * VC);
    g.cg_sdesc = c.take<u32>(A * VC);
    g.cg_scratch = c.take<u32>(A * static_cast<size_t>(g.cg_sc) * 32);
    g.cg_pool = c.take<u8>(A * static_cast<size_t>(g.pool_cap));
  };

  if (!nested) ctx->stats[2] += static_cast<unsigned long long>(n);
  // Pass 0 runs every window with the planned capacities.  Windows that come back flagged TABLE_OVERFLOW (graph larger
  // than the node array, walk search larger than the arena: the reference has no such limits) are re-assembled from
  // scratch by up to two more passes with 4x / 16x the node capacity and 8x / 64x the search arena -- only they are active, the workspace
  // is re-carved for fewer windows in flight.  What is still flagged afterwards is a limit growing cannot lift (more
  // than 16 edges on a node, a traversal cap the folded search cannot place: DESIGN.md section 7).
  for (int pass = 0; pass < 3; ++pass) {
    u32 const grow = pass == 0 ? 1u : (pass == 1 ? 4u : 16u);
    int const tc_log2_alloc = pass == 0 ? tc_first : tc_full;
    ws.mm_force_hbm = pass > 0 ? 1u : 0u;  // (a window whose LDS mate-mer set filled up comes back here: build.hip, k_mm_lds)
    {
      size_t const full = ((size_t(12) << mc_log2_alloc) + 255) & ~size_t(255);
      mm_share = (b.read_hint == nullptr || pass > 0) ? full : std::min<size_t>(full, size_t(256) << 10);
      if (const char* e = pass == 0 ? getenv("MA_MM_POOL_KB") : nullptr) mm_share = static_cast<size_t>(atoi(e)) << 10;  // tests: a pool that runs out
    }
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
    // (0.60 of the device for this stage AND the POA stage, which works in the same arena afterwards -- poa.hip; until round 6
    //  each had 0.30 of its own and a lane's 4096 windows of the headline workload went through both in two chunks)
    size_t budget = stage_budget(0.60, ctx->ws_build.cap, size_t(48) << 30, ctx->hbm_share);
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
        if (!nested && kpass >= 1 && ws.n_active > 0 && static_cast<u32>(ws.n_active) <= std::min(kSpecPending, kSpecPendingCap) && P.max_k > P.min_k &&
            !getenv("MA_NO_SPEC")) {
          // few windows are left on the ladder: their next rungs at once (speculate_tail), until none is pending
          u32 still = 0;
          MA_TRY_RC(speculate_tail(ctx, b, out, gate_approx, active, static_cast<u32>(ws.n_active), win_flags, counters + 8, &still));
          continue;  // (the next k_select_active moves what is still pending to its next rung, or finds the ladder exhausted)
        }
        if (ws.n_active > 0) {
          if (!nested) ctx->stats[3] += static_cast<unsigned long long>(ws.n_active);
          // (a single rung: every window is attempted at the k the capacity planning above counted it at -- the per-sequence
          //  instance bases and the windows' totals are already there.  INVARIANT the skip rests on: the planning
          //  call counted EVERY window at min_k == max_k == this k; seq_inst_base / win_ninst live in the misc carve that no
          //  chunk, capacity-retry pass or k_reset_overflowed writes (those reset node / table / arena state only).  Whoever
          //  changes win_k handling or moves those arrays must drop the skip; MA_RECOUNT_INST=1 recounts on every pass)
          if (nested || P.min_k != P.max_k || getenv("MA_RECOUNT_INST")) MA_TRY_RC(run_count_inst(ctx, b, ws, win0, nwin, counters));
          ws.tc_log2 = tc_log2_alloc;
          ws.mc_log2 = mc_log2_alloc;  // run_build_pass shrinks both to what this attempt needs
          MA_TRY_RC(run_build_pass(ctx, b, ws, counters + 12, tc_log2_alloc));
          if (ctx->collect) {
            unsigned long long* acc = nullptr;
            MA_TRY_RC(ma_dev_stats(ctx, &acc));
            hipLaunchKernelGGL(k_workload_stats, dim3(ws.n_active), dim3(256), 0, ctx->stream, ws, acc);
          }
          MA_TRY_RC(run_clean_pass(ctx, b, ws, out));
        }
      }
    }
  }
  return MA_OK;
}

}  // namespace ma
