// Internal definitions of the MI355X microassembly engine (product code; never includes oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/microasm.h"

namespace ma {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i8 = int8_t;
using i16 = int16_t;
using i32 = int32_t;
using i64 = int64_t;
using f64 = double;
using f32 = float;

// node-identity hash shared with the oracle (oracle/common.hpp: HashStr64)
constexpr u64 kHashP = 0x9E3779B97F4A7C15ULL;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// One staged copy of a lane's slice of a host batch (MA_MEM_HOST route of ma_process_batch): two per lane, so that
// ma_prefetch_batch can upload the next batch while this one computes.
struct InputSet {
  DevBuf bufs[10];
  std::vector<uint32_t> h_rwo;  // the lane's rebased read_win_off while its upload is in flight
};

struct KernelTimer {
  const char* name;
  hipEvent_t beg, end;
};

}  // namespace ma

struct ma_ctx {
  ma_params_t prm;
  int device = 0;
  int memspace = MA_MEM_HOST;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;  // created by ma_create (non-blocking); `stream` points here until ma_set_stream
  std::string err;     // last error: written through ma_set_err (worker threads of the host route write it too)
  std::mutex err_mu;
  // staging for MA_MEM_HOST
  ma::DevBuf in_stage[10];
  std::vector<ma::DevBuf> out_stage;
  // per-stage workspaces (grow-only, reused across calls)
  ma::DevBuf ws_build, ws_nodes, ws_clean, ws_aln, ws_misc, ws_gen, ws_mm;  // (ws_build: the assembly stage's arena, then the POA stage's)
  // speculative tail of the k ladder (assemble.hip: speculate_tail): the derived batch, its outputs and the bookkeeping of the
  // nested pass; spec_k = the rung every window of the nested pass is built at (null outside of it)
  ma::DevBuf spec_data, spec_out, spec_nodes;
  const uint32_t* spec_k = nullptr;
  // annotation tables (annotate.hip): rebuilt when the GC fraction or max_hap_len changes
  ma::DevBuf ws_cx;
  double cx_gc = -1.0;
  int cx_ml = 0;
  std::vector<double> cx_host;
  // kernel timing
  std::vector<ma::KernelTimer> timers;
  size_t timers_used = 0;
  bool timing = true;
  bool accumulate = false;
  unsigned long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // see ma_last_stats
  bool collect = false;          // ma_timing_control mode 3: also gather the workload statistics (device side)
  ma::DevBuf dev_stats;          // [kDevStats] u64: [0..5] k_workload_stats, [6] DP cells of the read aligner's regions,
                                 // [7] POA band cells, [8] band fills, [9] cells of full fills, [10] POA alignments, [11] closed-form ones
  bool dev_stats_clean = false;
  // ma_process_batch splits a batch into `n_lanes` contiguous window ranges that run concurrently on child
  // contexts (own stream + workspaces): the stages have complementary bottlenecks (latency-bound graph
  // cleaning, VALU-bound DP, HBM-bound table passes), so two batches in flight fill the gaps of one.
  int n_lanes = 0;  // 0 = automatic (3 for batches of >= 6144 windows, 2 for >= 2048, else 1)
  std::vector<ma_ctx*> lanes;
  ma::DevBuf lane_rwo;      // rebased read_win_off of a child lane
  hipEvent_t lane_done = nullptr;
  hipStream_t hi_stream = nullptr;  // a lane's POA rounds run here, at the greatest stream priority (poa.hip: launch_msa)
  hipEvent_t hi_ev = nullptr;
  bool hi_failed = false;
  double hbm_share = 1.0;   // fraction of the device this context plans its workspaces for
  hipEvent_t sync_ev = nullptr;  // blocking-sync event: host threads sleep instead of spinning while the stream drains
  // host route (MA_MEM_HOST) of ma_process_batch, per lane: packed result records (pack.hip) and their pinned landing area
  ma::DevBuf pack_aux;
  void* pin[2] = {nullptr, nullptr};      // one landing area per input set: a batch's records are read by the caller's
  size_t pin_cap[2] = {0, 0};             // thread while the lane already computes the next batch
  size_t last_packed = 0;                 // (lane) bytes of packed records of the lane's last batch: sizes the next landing area
  void* host_async = nullptr;             // (parent) worker threads + jobs of the host route (api.hip: HostAsync)
  void* dev_pool = nullptr;               // (parent) the lanes' worker threads of the device route (api.hip: DevLanePool)
  hipStream_t copy_stream2 = nullptr;     // the pieces of an upload alternate between the two copy streams (api.hip: run_copy_ops)
  ma::InputSet in_sets[2];       // (lane) input staging, double buffered
  // (parent) ma_prefetch_batch: which batch each set of the lanes holds (null: free), in which order they were filled,
  // and the stream the uploads of a prefetch run on (api.hip: uploader thread)
  const ma_batch_t* pf_batch[2] = {nullptr, nullptr};
  int64_t pf_sig[2][3] = {{0, 0, 0}, {0, 0, 0}};  // n_windows, n_reads, lanes of the prefetched batch
  unsigned long long pf_seq[2] = {0, 0}, pf_counter = 0;
  hipStream_t copy_stream = nullptr;

  void tic(const char* name);
  void toc();
};

// the error string of a context is shared by the caller's thread, the uploader and the lane workers of the host route
inline void ma_set_err(ma_ctx* c, std::string m) {
  std::lock_guard<std::mutex> g(c->err_mu);
  c->err = std::move(m);
}
inline std::string ma_get_err(ma_ctx* c) {
  std::lock_guard<std::mutex> g(c->err_mu);
  return c->err;
}

#define MA_HIP(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess) {                                                            \
      ma_set_err((ctx), std::string(#call) + ": " + hipGetErrorString(_e));             \
      return MA_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

// ma_timing_control mode 3: the device-side statistics block (zeroed once per statistics region); null when not collecting
constexpr int kDevStats = 16;
inline int ma_dev_stats(ma_ctx* ctx, unsigned long long** out) {
  *out = nullptr;
  if (!ctx->collect) return MA_OK;
  MA_HIP(ctx, ctx->dev_stats.reserve(sizeof(unsigned long long) * kDevStats));
  if (!ctx->dev_stats_clean) {
    MA_HIP(ctx, hipMemsetAsync(ctx->dev_stats.p, 0, sizeof(unsigned long long) * kDevStats, ctx->stream));
    ctx->dev_stats_clean = true;
  }
  *out = ctx->dev_stats.as<unsigned long long>();
  return MA_OK;
}

// Wait for the context's stream without burning a host core: one process per GPU and up to three lane threads per
// process would otherwise spin on hipStreamSynchronize (8 GPUs: 24 busy cores for nothing).
inline hipError_t ma_stream_sync(ma_ctx* ctx) {
  if (getenv("MA_SPIN_SYNC")) return hipStreamSynchronize(ctx->stream);
  if (!ctx->sync_ev) {
    hipError_t const e = hipEventCreateWithFlags(&ctx->sync_ev, hipEventBlockingSync | hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  hipError_t const e = hipEventRecord(ctx->sync_ev, ctx->stream);
  if (e != hipSuccess) return e;
  return hipEventSynchronize(ctx->sync_ev);
}

#define MA_TRY_RC(expr)            \
  do {                             \
    int _rc = (expr);              \
    if (_rc != MA_OK) return _rc;  \
  } while (0)

namespace ma {

// device-side views (all pointers are device pointers)
struct DBatch {
  int n_windows;
  i64 n_reads;
  const u8* ref_bases;
  const u32* ref_off;
  const u32* read_win_off;
  const u64* read_off;
  const u8* read_bases;
  const u8* read_quals;
  const u32* read_qname_id;
  const u8* read_sample;
  const u8* read_flags;
  const i32* read_hint;  // may be null
};

int launch_gate(ma_ctx* ctx, const DBatch& b, u32* max_approx, u32* max_exact);
int launch_assemble(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& o, const u32* gate_approx);
int launch_msa(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o);
int launch_genotype(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& v,
                    const ma_geno_out_t& o);
int launch_annotate(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& v, double gc_frac,
                    const ma_cx_out_t& o);

// Workspace budget of one stage: a fixed share of the device's HBM (288 GB on MI355X), so that the
// chunking does not depend on the order in which the stages first allocated, capped by what is free now
// (the stage's own buffer counts as free: it is reused).
inline size_t stage_budget(double share_of_total, size_t own_cap, size_t fallback, double ctx_share = 1.0) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return fallback;
  // several engines on one GPU (one per stream): MA_HBM_SHARE = the fraction of the device this one may plan for
  double scale = ctx_share;
  if (const char* e = getenv("MA_HBM_SHARE")) {
    double const v = atof(e);
    if (v > 0.0 && v <= 1.0) scale *= v;
  }
  size_t const want = static_cast<size_t>(static_cast<double>(total_b) * share_of_total * scale);
  size_t const avail = free_b + own_cap;
  size_t const guard = size_t(2) << 30;
  size_t const cap = avail > guard ? avail - guard : avail / 2;
  return want < cap ? want : cap;
}

// atomicMax on a batch-wide maximum that thousands of workgroups report to (sizes a later kernel's LDS, a table stride): the
// read-modify-writes of one address are served one after the other by the L2 -- 2 M of them in k_count_inst were 0.3 ms of
// its 0.38 ms.  A relaxed load first: a value that cannot raise the maximum (almost all, once a few have landed) costs a
// read that many wavefronts share; a stale read only costs the atomic it would have cost anyway.
__device__ __forceinline__ void atomic_max_lazy(unsigned int* p, unsigned int v) {
  if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
}

// Branch-free on purpose: a `switch` over the base compiles to a tree of divergent branches per byte.
//   'A' 0x41  'C' 0x43  'G' 0x47  'T' 0x54: bit 1 separates {A,T} from {C,G}; A^T = 0x15, C^G = 0x04.
constexpr u32 kAcgtBits = (1u << ('A' - 'A')) | (1u << ('C' - 'A')) | (1u << ('G' - 'A')) | (1u << ('T' - 'A'));

__device__ __forceinline__ bool dev_is_acgt_upper(u32 u) {  // u already upper-cased (c & 0xDF)
  u32 const k = u - 'A';
  return k < 32u && ((kAcgtBits >> (k & 31u)) & 1u);
}

__device__ __forceinline__ u8 dev_complement(u8 b) {  // base/rev_comp.h:15-31
  u32 const c = b, u = c & 0xDFu, lower = c & 0x20u;
  u32 const comp = (u ^ ((u & 2u) ? 0x04u : 0x15u)) | lower;
  bool const acgt = dev_is_acgt_upper(u);  // then c is that letter, upper or lower case
  u32 const other = (c == 'n') ? u32('n') : u32('N');
  return static_cast<u8>(acgt ? comp : other);
}

__device__ __forceinline__ u32 enc_base(u8 b) {  // scoring_constants.h:48-74: A 0, C 1, G 2, T 3, other 4
  u32 const c = b;
  u32 e = (c >> 1) & 3u;  // A 0, C 1, T 2, G 3
  e ^= e >> 1;            // A 0, C 1, G 2, T 3
  return dev_is_acgt_upper(c & 0xDFu) ? e : 4u;
}

}  // namespace ma
