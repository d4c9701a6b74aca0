// C-ABI entry points of libmicroasm.so (see include/microasm.h).  No CPU fallback: every entry
// point fails with MA_ERR_NO_DEVICE when there is no HIP device.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>

#include "ma_internal.h"
#include "pack.h"

using namespace ma;

void ma_ctx::tic(const char* name) {
  if (!timing) return;
  if (timers_used == timers.size()) {
    KernelTimer t{name, nullptr, nullptr};
    (void)hipEventCreate(&t.beg);
    (void)hipEventCreate(&t.end);
    timers.push_back(t);
  }
  timers[timers_used].name = name;
  (void)hipEventRecord(timers[timers_used].beg, stream);
}
void ma_ctx::toc() {
  if (!timing) return;
  (void)hipEventRecord(timers[timers_used].end, stream);
  timers_used++;
}

namespace {

struct OutField {
  size_t offset;   // byte offset of the pointer member inside the out struct
  size_t bytes;    // payload size
};

template <class S, class T>
size_t off_of(T* S::*m) {
  return reinterpret_cast<size_t>(&(reinterpret_cast<S*>(0)->*m));
}

std::vector<OutField> gate_fields(const ma_params_t&, int n) {
  return {{off_of(&ma_gate_out_t::max_approx), 4ull * n}, {off_of(&ma_gate_out_t::max_exact), 4ull * n}};
}
std::vector<OutField> asm_fields(const ma_params_t& p, int n) {
  size_t const N = n, MC = p.max_comps, MH = p.max_haps, ML = p.max_hap_len, MR = p.max_runs;
  return {{off_of(&ma_asm_out_t::win_status), 4 * N}, {off_of(&ma_asm_out_t::win_k), 4 * N},
          {off_of(&ma_asm_out_t::win_ncomp), 4 * N}, {off_of(&ma_asm_out_t::comp_anchor), 4 * N * MC},
          {off_of(&ma_asm_out_t::comp_hap0), 4 * N * MC}, {off_of(&ma_asm_out_t::comp_nhaps), 4 * N * MC},
          {off_of(&ma_asm_out_t::comp_cx), 4 * N * MC * 3}, {off_of(&ma_asm_out_t::comp_cxf), 8 * N * MC * 4},
          {off_of(&ma_asm_out_t::hap_len), 4 * N * MH}, {off_of(&ma_asm_out_t::hap_nruns), 4 * N * MH},
          {off_of(&ma_asm_out_t::hap_stats), 8 * N * MH * 6}, {off_of(&ma_asm_out_t::hap_bases), N * MH * ML},
          {off_of(&ma_asm_out_t::hap_runs), 4 * N * MH * MR * 2}};
}
std::vector<OutField> var_fields(const ma_params_t& p, int n) {
  size_t const N = n, MH = p.max_haps, MV = p.max_vars, MA = p.max_alts, MP = p.max_allele_bytes;
  return {{off_of(&ma_var_out_t::win_nvars), 4 * N}, {off_of(&ma_var_out_t::var_comp), 4 * N * MV},
          {off_of(&ma_var_out_t::var_pos), 4 * N * MV}, {off_of(&ma_var_out_t::var_ref_start), 4 * N * MV},
          {off_of(&ma_var_out_t::var_ref_off), 4 * N * MV}, {off_of(&ma_var_out_t::var_ref_len), 4 * N * MV},
          {off_of(&ma_var_out_t::var_nalts), 4 * N * MV}, {off_of(&ma_var_out_t::alt_off), 4 * N * MV * MA},
          {off_of(&ma_var_out_t::alt_len), 4 * N * MV * MA}, {off_of(&ma_var_out_t::alt_type), 4 * N * MV * MA},
          {off_of(&ma_var_out_t::alt_length), 4 * N * MV * MA},
          {off_of(&ma_var_out_t::var_hap_allele), N * MV * MH}, {off_of(&ma_var_out_t::var_hap_start), 4 * N * MV * MH},
          {off_of(&ma_var_out_t::allele_pool), N * MP}};
}
std::vector<OutField> geno_fields(const ma_params_t& p, int n, i64 nr) {
  size_t const N = n, R = static_cast<size_t>(nr), MH = p.max_haps, MV = p.max_vars, MA = p.max_alts,
               S = p.num_samples, MCG = p.max_cigar;
  return {{off_of(&ma_geno_out_t::allele_counts), 4 * N * MV * S * (MA + 1) * 2},
          {off_of(&ma_geno_out_t::var_qual), 8 * N * MV},
          {off_of(&ma_geno_out_t::aln_rec), 4 * R * MH * 6},
          {off_of(&ma_geno_out_t::aln_cigar), 4 * R * MH * (1 + MCG)},
          {off_of(&ma_geno_out_t::asg_allele), R * MV},
          {off_of(&ma_geno_out_t::asg_score), 8 * R * MV},
          {off_of(&ma_geno_out_t::var_pl), 4 * N * MV * S * ((MA + 1) * (MA + 2) / 2)},
          {off_of(&ma_geno_out_t::var_gq), 4 * N * MV * S}};
}

std::vector<OutField> cx_fields(const ma_params_t& p, int n) {
  size_t const N = n, MV = p.max_vars;
  return {{off_of(&ma_cx_out_t::seq_cx_i), 4 * N * MV * 4}, {off_of(&ma_cx_out_t::seq_cx_f), 4 * N * MV * 4},
          {off_of(&ma_cx_out_t::seq_cx_d), 8 * N * MV * 3}, {off_of(&ma_cx_out_t::graph_cx), 8 * N * MV * 3}};
}

void*& ptr_at(void* strct, size_t off) { return *reinterpret_cast<void**>(static_cast<char*>(strct) + off); }
void* ptr_at(const void* strct, size_t off) {
  return *reinterpret_cast<void* const*>(static_cast<const char*>(strct) + off);
}

// Device mirror of a caller output struct.  MA_MEM_DEVICE: aliases the caller's pointers.
// MA_MEM_HOST: one device allocation per non-NULL member; download() copies results back.
template <class S>
struct OutMirror {
  S dev;
  const S* host = nullptr;
  std::vector<OutField> fields;
  std::vector<DevBuf*> bufs;
  int prepare(ma_ctx* ctx, const S* user, std::vector<OutField> f, size_t stage_base, bool upload) {
    host = user;
    fields = std::move(f);
    std::memset(&dev, 0, sizeof(dev));
    if (!user) return MA_OK;
    if (ctx->memspace == MA_MEM_DEVICE) {
      dev = *user;
      return MA_OK;
    }
    if (ctx->out_stage.size() < stage_base + fields.size()) ctx->out_stage.resize(stage_base + fields.size());
    for (size_t i = 0; i < fields.size(); ++i) {
      void* hp = ptr_at(user, fields[i].offset);
      if (!hp) continue;
      DevBuf& b = ctx->out_stage[stage_base + i];
      MA_HIP(ctx, b.reserve(fields[i].bytes));
      ptr_at(&dev, fields[i].offset) = b.p;
      if (upload) MA_HIP(ctx, hipMemcpyAsync(b.p, hp, fields[i].bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    return MA_OK;
  }
  int download(ma_ctx* ctx) {
    if (!host || ctx->memspace == MA_MEM_DEVICE) return MA_OK;
    for (auto const& f : fields) {
      void* hp = ptr_at(host, f.offset);
      void* dp = ptr_at(&dev, f.offset);
      if (!hp || !dp) continue;
      MA_HIP(ctx, hipMemcpyAsync(hp, dp, f.bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    return MA_OK;
  }
};

int stage_batch(ma_ctx* ctx, const ma_batch_t* b, DBatch* d) {
  if (!b || b->n_windows < 0 || b->n_reads < 0) return MA_ERR_ARG;
  d->n_windows = b->n_windows;
  d->n_reads = b->n_reads;
  if (ctx->memspace == MA_MEM_DEVICE) {
    d->ref_bases = b->ref_bases; d->ref_off = b->ref_off; d->read_win_off = b->read_win_off;
    d->read_off = b->read_off; d->read_bases = b->read_bases; d->read_quals = b->read_quals;
    d->read_qname_id = b->read_qname_id; d->read_sample = b->read_sample; d->read_flags = b->read_flags;
    d->read_hint = b->read_hint;
    return MA_OK;
  }
  size_t const n = b->n_windows, nr = static_cast<size_t>(b->n_reads);
  size_t const ref_bytes = n ? b->ref_off[n] : 0;
  size_t const read_bytes = nr ? b->read_off[nr] : 0;
  struct Item { const void* src; size_t bytes; size_t pad; const void** dst; };
  d->read_hint = nullptr;
  Item items[10] = {
      {b->ref_bases, ref_bytes, 64, reinterpret_cast<const void**>(&d->ref_bases)},
      {b->ref_off, 4 * (n + 1), 0, reinterpret_cast<const void**>(&d->ref_off)},
      {b->read_win_off, 4 * (n + 1), 0, reinterpret_cast<const void**>(&d->read_win_off)},
      {b->read_off, 8 * (nr + 1), 0, reinterpret_cast<const void**>(&d->read_off)},
      {b->read_bases, read_bytes, 64, reinterpret_cast<const void**>(&d->read_bases)},
      {b->read_quals, read_bytes, 64, reinterpret_cast<const void**>(&d->read_quals)},
      {b->read_qname_id, 4 * nr, 0, reinterpret_cast<const void**>(&d->read_qname_id)},
      {b->read_sample, nr, 0, reinterpret_cast<const void**>(&d->read_sample)},
      {b->read_flags, nr, 0, reinterpret_cast<const void**>(&d->read_flags)},
      {b->read_hint, b->read_hint ? 4 * nr : 0, 0, reinterpret_cast<const void**>(&d->read_hint)}};
  for (int i = 0; i < 10; ++i) {
    if (i == 9 && !b->read_hint) break;
    MA_HIP(ctx, ctx->in_stage[i].reserve(items[i].bytes + items[i].pad + 16));
    if (items[i].bytes)
      MA_HIP(ctx, hipMemcpyAsync(ctx->in_stage[i].p, items[i].src, items[i].bytes, hipMemcpyHostToDevice, ctx->stream));
    if (items[i].pad)
      MA_HIP(ctx, hipMemsetAsync(static_cast<char*>(ctx->in_stage[i].p) + items[i].bytes, 0, items[i].pad, ctx->stream));
    *items[i].dst = ctx->in_stage[i].p;
  }
  return MA_OK;
}

int check_params(const ma_params_t& p) {
  if (p.min_k < 5 || p.max_k > 255 || p.min_k > p.max_k || p.k_step <= 0) return MA_ERR_PARAM;
  if ((p.min_k & 1) == 0 || (p.k_step & 1) != 0) return MA_ERR_PARAM;  // k must stay odd (graph_params.h:11-26)
  if (p.num_samples < 1 || p.num_samples > 8) return MA_ERR_PARAM;
  if (p.max_mismatch < 0 || p.max_mismatch > 3) return MA_ERR_PARAM;
  if (p.max_comps < 1 || p.max_comps > 16 || p.max_haps < 2 || p.max_haps > 32) return MA_ERR_PARAM;
  if (p.max_hap_len < 256 || p.max_hap_len > 8192 || p.max_runs < 8) return MA_ERR_PARAM;
  if (p.max_vars < 1 || p.max_alts < 1 || p.max_alts > 15 || p.max_allele_bytes < 64) return MA_ERR_PARAM;
  if (p.aln_tier < 0 || p.aln_tier > 3 || p.max_cigar < 4 || p.max_cigar > 64) return MA_ERR_PARAM;
  if (p.min_aln_score < 1) return MA_ERR_PARAM;  // row-0 end cells are never hits (align.hip)
  if (p.bfs_limit < 1) return MA_ERR_PARAM;
  return MA_OK;
}

}  // namespace


namespace {

__global__ void k_rebase_u32(const u32* src, u32* dst, int n, u32 base) {
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] - base;
}

// advance the non-null pointer members [first, last) of an output struct by `units` x (bytes per unit)
template <class S>
void advance_fields(S* s, const std::vector<OutField>& per_unit, size_t first, size_t last, size_t units) {
  for (size_t i = first; i < last && i < per_unit.size(); ++i) {
    void*& p = ptr_at(s, per_unit[i].offset);
    if (p) p = static_cast<char*>(p) + per_unit[i].bytes * units;
  }
}

// One contiguous window range [w0, w1) of the batch on a child context.  All pointers are device pointers.
int run_lane(ma_ctx* ch, hipEvent_t start, const DBatch& full, int w0, int w1, u32 r0, u32 r1, ma_gate_out_t g,
             ma_asm_out_t a, ma_var_out_t v, ma_geno_out_t q, int lane_index) {
  MA_HIP(ch, hipSetDevice(ch->device));
  MA_HIP(ch, hipStreamWaitEvent(ch->stream, start, 0));
  (void)lane_index;  // (starting the lanes a few ms apart was measured: every ms of stagger is lost, the lanes do overlap)
  int const n = w1 - w0;
  DBatch d = full;
  d.n_windows = n;
  d.n_reads = static_cast<i64>(r1) - r0;
  d.ref_off = full.ref_off + w0;                 // absolute offsets into ref_bases
  MA_HIP(ch, ch->lane_rwo.reserve(4ull * (n + 1) + 64));
  u32* rwo = ch->lane_rwo.as<u32>();
  hipLaunchKernelGGL(k_rebase_u32, dim3((n + 1 + 255) / 256), dim3(256), 0, ch->stream, full.read_win_off + w0, rwo,
                     n + 1, r0);
  d.read_win_off = rwo;                          // the lane's reads are numbered from 0 ...
  d.read_off = full.read_off + r0;               // ... their byte offsets stay absolute
  d.read_qname_id = full.read_qname_id + r0;
  d.read_sample = full.read_sample + r0;
  d.read_flags = full.read_flags + r0;
  d.read_hint = full.read_hint ? full.read_hint + r0 : nullptr;
  ma_params_t const& p = ch->prm;
  advance_fields(&g, gate_fields(p, 1), 0, 2, w0);
  advance_fields(&a, asm_fields(p, 1), 0, 99, w0);
  advance_fields(&v, var_fields(p, 1), 0, 99, w0);
  std::vector<OutField> const gf = geno_fields(p, 1, 1);
  advance_fields(&q, gf, 0, 2, w0);   // allele_counts, var_qual: per window
  advance_fields(&q, gf, 2, 6, r0);   // alignment / assignment taps: per read
  advance_fields(&q, gf, 6, 8, w0);   // PL, GQ: per window
  MA_TRY_RC(launch_gate(ch, d, g.max_approx, g.max_exact));
  MA_TRY_RC(launch_assemble(ch, d, a, g.max_approx));
  MA_TRY_RC(launch_msa(ch, d, a, v));
  MA_TRY_RC(launch_genotype(ch, d, a, v, q));
  MA_HIP(ch, hipEventRecord(ch->lane_done, ch->stream));
  return MA_OK;
}

// The lanes of the device route run on PERSISTENT threads (one per lane, parked on a condition variable between calls): a call
// hands them its closure and waits for the last of them -- no thread is created or joined per call (round 3 did both, 4 x ~40 us
// and a first-touch of the HIP runtime's per-thread state on every call).
struct DevLanePool {
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> th;
  std::function<void(int)> fn;
  unsigned long long gen = 0;
  int active = 0, pending = 0;
  bool stop = false;
  void ensure(int n) {
    while (static_cast<int>(th.size()) < n) {
      int const k = static_cast<int>(th.size());
      th.emplace_back([this, k] {
        unsigned long long seen = 0;
        for (;;) {
          std::function<void(int)> f;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return stop || (gen != seen && k < active); });
            if (stop) return;
            seen = gen;
            f = fn;
          }
          f(k);
          {
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
          }
        }
      });
    }
  }
  void run(int n, std::function<void(int)> f) {
    ensure(n);
    {
      std::lock_guard<std::mutex> lk(mu);
      fn = std::move(f);
      active = n;
      pending = n;
      ++gen;
    }
    cv_go.notify_all();
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~DevLanePool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_go.notify_all();
    for (auto& t : th) t.join();
  }
};

int process_in_lanes(ma_ctx* ctx, int lanes, const DBatch& d, const ma_gate_out_t& g, const ma_asm_out_t& a,
                     const ma_var_out_t& v, const ma_geno_out_t& q) {
  while (static_cast<int>(ctx->lanes.size()) < lanes) {
    ma_ctx* ch = new (std::nothrow) ma_ctx();
    if (!ch) return MA_ERR_NOMEM;
    ch->device = ctx->device;
    ch->memspace = MA_MEM_DEVICE;
    MA_HIP(ctx, hipStreamCreateWithFlags(&ch->stream, hipStreamNonBlocking));
    MA_HIP(ctx, hipEventCreateWithFlags(&ch->lane_done, hipEventDisableTiming));
    ctx->lanes.push_back(ch);
  }
  if (!ctx->lane_done) MA_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_done, hipEventDisableTiming));
  // window boundaries and the read index at each of them
  std::vector<int> wb(lanes + 1);
  std::vector<u32> rb(lanes + 1, 0);
  for (int k = 0; k <= lanes; ++k) wb[k] = static_cast<int>(static_cast<long long>(d.n_windows) * k / lanes);
  for (int k = 0; k <= lanes; ++k)
    MA_HIP(ctx, hipMemcpyAsync(&rb[k], d.read_win_off + wb[k], 4, hipMemcpyDeviceToHost, ctx->stream));
  MA_HIP(ctx, hipEventRecord(ctx->lane_done, ctx->stream));  // inputs (and staging copies) are ready after this
  MA_HIP(ctx, ma_stream_sync(ctx));
  std::vector<int> rc(lanes, MA_OK);
  for (int k = 0; k < lanes; ++k) {
    ma_ctx* ch = ctx->lanes[k];
    ch->prm = ctx->prm;
    ch->timing = ctx->timing;
    ch->accumulate = ctx->accumulate;
    ch->collect = ctx->collect;
    if (!ch->accumulate) ch->timers_used = 0;
    ch->hbm_share = ctx->hbm_share / lanes;
  }
  if (!ctx->dev_pool) ctx->dev_pool = new DevLanePool();
  static_cast<DevLanePool*>(ctx->dev_pool)->run(lanes, [&](int k) {
    rc[k] = run_lane(ctx->lanes[k], ctx->lane_done, d, wb[k], wb[k + 1], rb[k], rb[k + 1], g, a, v, q, k);
  });
  for (int k = 0; k < lanes; ++k) {
    if (rc[k] != MA_OK) {
      ma_set_err(ctx, "lane " + std::to_string(k) + ": " + ma_get_err(ctx->lanes[k]));
      return rc[k];
    }
    MA_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->lanes[k]->lane_done, 0));
  }
  return MA_OK;
}

// ---- the host route (MA_MEM_HOST) of ma_process_batch -------------------------------------------------------------------
// Every lane stages ITS slice of the caller's input arrays (or finds it staged by ma_prefetch_batch), runs the four stages,
// packs the used part of its results (pack.hip) and copies the records to a pinned landing area; the caller's thread scatters
// them into the caller's arrays.  What crosses PCIe on the way back is what the engine wrote: ~5 KB instead of 95 KB per
// window.
//
// The lanes are PERSISTENT worker threads with a job queue each.  ma_prefetch_batch does not only upload the next batch: it
// also queues the batch's compute jobs, so a lane goes from its last kernel of batch j straight to the first kernel of batch
// j + 1 -- the lanes drift apart instead of starting and ending every batch in lockstep, and the end-of-batch work of one
// (pack, download, the caller's scatter, the call returning and the next one coming in) runs under the kernels of the
// others.  ma_process_batch(j + 1) then only waits for the records and scatters them.  A job that was queued ahead assumes
// that the call will ask for the same optional outputs as the last one did and run under the same parameters; if it does
// not, the job's results are dropped and the batch is computed again, in the call.
struct LaneOut {
  ma_gate_out_t g{};
  ma_asm_out_t a{};
  ma_var_out_t v{};
  ma_geno_out_t q{};
};

int ensure_pinned(ma_ctx* ch, int set, size_t bytes) {
  if (bytes <= ch->pin_cap[set]) return MA_OK;
  if (ch->pin[set]) (void)hipHostFree(ch->pin[set]);
  ch->pin[set] = nullptr;
  ch->pin_cap[set] = 0;
  size_t const want = bytes + bytes / 4 + 4096;
  MA_HIP(ch, hipHostMalloc(&ch->pin[set], want, hipHostMallocDefault));
  ch->pin_cap[set] = want;
  return MA_OK;
}

template <class S>
int alloc_fields(ma_ctx* ch, S* dev, const std::vector<OutField>& f, size_t stage_base, const std::vector<bool>& want) {
  std::memset(dev, 0, sizeof(*dev));
  if (ch->out_stage.size() < stage_base + f.size()) ch->out_stage.resize(stage_base + f.size());
  for (size_t i = 0; i < f.size(); ++i) {
    if (!want[i]) continue;  // an optional member the caller left out
    DevBuf& b = ch->out_stage[stage_base + i];
    MA_HIP(ch, b.reserve(f[i].bytes + 16));
    ptr_at(dev, f[i].offset) = b.p;
  }
  return MA_OK;
}

// The output arrays of the route, as tables: which struct (0 gate, 1 asm, 2 var, 3 geno) and member, bytes per window.
struct DenseDesc { int strct; size_t off; size_t win_bytes; };  // small per-window arrays: copied whole
struct SegDesc { int strct; size_t off; size_t win_stride, unit; u32 kind; };  // packed: only what a window uses
std::vector<DenseDesc> dense_table(const ma_params_t& p) {
  size_t const MC = p.max_comps, MH = p.max_haps;
  return {{0, off_of(&ma_gate_out_t::max_approx), 4}, {0, off_of(&ma_gate_out_t::max_exact), 4},
          {1, off_of(&ma_asm_out_t::win_status), 4}, {1, off_of(&ma_asm_out_t::win_k), 4},
          {1, off_of(&ma_asm_out_t::win_ncomp), 4}, {1, off_of(&ma_asm_out_t::comp_anchor), 4 * MC},
          {1, off_of(&ma_asm_out_t::comp_hap0), 4 * MC}, {1, off_of(&ma_asm_out_t::comp_nhaps), 4 * MC},
          {1, off_of(&ma_asm_out_t::comp_cx), 12 * MC}, {1, off_of(&ma_asm_out_t::comp_cxf), 32 * MC},
          {1, off_of(&ma_asm_out_t::hap_len), 4 * MH}, {1, off_of(&ma_asm_out_t::hap_nruns), 4 * MH},
          {1, off_of(&ma_asm_out_t::hap_stats), 48 * MH}, {2, off_of(&ma_var_out_t::win_nvars), 4}};
}
std::vector<SegDesc> seg_table(const ma_params_t& p) {
  size_t const MH = p.max_haps, MV = p.max_vars, MA_ = p.max_alts, S = p.num_samples, ML = p.max_hap_len, MR = p.max_runs,
               MP = p.max_allele_bytes;
  size_t const G = (MA_ + 1) * (MA_ + 2) / 2;
  auto var = [&](int strct, size_t off, size_t elem, size_t per_var) {
    return SegDesc{strct, off, MV * per_var * elem, per_var * elem, PK_VAR};
  };
  return {{1, off_of(&ma_asm_out_t::hap_bases), MH * ML, ML, PK_HAP_BASES},
          {1, off_of(&ma_asm_out_t::hap_runs), MH * MR * 8, MR * 8, PK_HAP_RUNS},
          var(2, off_of(&ma_var_out_t::var_comp), 4, 1), var(2, off_of(&ma_var_out_t::var_pos), 4, 1),
          var(2, off_of(&ma_var_out_t::var_ref_start), 4, 1), var(2, off_of(&ma_var_out_t::var_ref_off), 4, 1),
          var(2, off_of(&ma_var_out_t::var_ref_len), 4, 1), var(2, off_of(&ma_var_out_t::var_nalts), 4, 1),
          var(2, off_of(&ma_var_out_t::alt_off), 4, MA_), var(2, off_of(&ma_var_out_t::alt_len), 4, MA_),
          var(2, off_of(&ma_var_out_t::alt_type), 4, MA_), var(2, off_of(&ma_var_out_t::alt_length), 4, MA_),
          var(2, off_of(&ma_var_out_t::var_hap_allele), 1, MH), var(2, off_of(&ma_var_out_t::var_hap_start), 4, MH),
          {2, off_of(&ma_var_out_t::allele_pool), MP, 0, PK_POOL},
          var(3, off_of(&ma_geno_out_t::allele_counts), 4, S * (MA_ + 1) * 2), var(3, off_of(&ma_geno_out_t::var_qual), 8, 1),
          var(3, off_of(&ma_geno_out_t::var_pl), 4, S * G), var(3, off_of(&ma_geno_out_t::var_gq), 4, S)};
}
struct OutPtrs {  // the four output structs of a call, by table index
  const void* s[4];
  void* at(int strct, size_t off) const { return s[strct] ? ptr_at(s[strct], off) : nullptr; }
};
// which packed arrays the caller asked for (bit i = entry i of seg_table)
u32 seg_mask_of(const ma_params_t& p, OutPtrs const& u) {
  std::vector<SegDesc> const t = seg_table(p);
  u32 m = 0;
  for (size_t i = 0; i < t.size(); ++i)
    if (u.at(t[i].strct, t[i].off)) m |= 1u << i;
  return m;
}
bool wants_taps(const ma_geno_out_t* q) { return q->aln_rec || q->aln_cigar || q->asg_allele || q->asg_score; }

// Uploads run in PIECES with at most two of them queued: the copy engine serves its ring in order, and the lanes that
// are computing meanwhile keep sending it small transfers of their own (chunk plans, work lists: dozens per batch) that wait
// behind everything queued before them.  Measured with a background copy beside the device-resident route: 1.6 GB queued
// as one burst (in whatever piece size) costs the kernels' side +20 %, single 256 MB transfers 15x, 16 MB pieces each waited
// for before the next is queued +2 % at 93 % of the link's rate.
double wall_ms() {  // (MA_VERBOSE timelines)
  static auto const t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
struct CopyOp {
  void* dst;
  const void* src;  // null: zero `bytes` bytes at dst
  size_t bytes;
};
template <class F>
int run_copy_ops(ma_ctx* owner, hipStream_t stream, hipStream_t stream2, std::vector<CopyOp> const& ops,
                 std::vector<size_t> const& group_end, F&& group_done) {
  // one piece at a time, waited for with a plain stream synchronise (an event or marker behind a DMA copy is a packet on a
  // compute queue and waits for that queue's turn: 28-33 GB/s instead of 55); MA_UPLOAD_TWO_STREAMS alternates the pieces
  // between two streams, two in flight (measured: 46.0 instead of 43.2 ms per batch of 8192 windows)
  static size_t const piece = (getenv("MA_UPLOAD_CHUNK_MB") ? static_cast<size_t>(atoi(getenv("MA_UPLOAD_CHUNK_MB"))) : 16u) << 20;
  auto const t_begin = std::chrono::steady_clock::now();
  hipStream_t const st[2] = {stream, stream2 ? stream2 : stream};
  size_t queued = 0;
  static int const dbg_skip = getenv("MA_DEBUG_SKIP_UPLOAD") ? atoi(getenv("MA_DEBUG_SKIP_UPLOAD")) : 0;  // developer experiment:
  static std::atomic<int> dbg_calls{0};                                                                                // the buffers keep what
  if (dbg_skip > 0 && ++dbg_calls > dbg_skip) {
    for (size_t g = 0; g < group_end.size(); ++g) group_done(g);
    return MA_OK;
  }                                                // the first uploads put there
  for (CopyOp const& op : ops)  // the pads first: a handful of tiny fills
    if (op.bytes && !op.src) MA_HIP(owner, hipMemsetAsync(op.dst, 0, op.bytes, stream));
  size_t grp = 0;
  for (size_t oi = 0; oi < ops.size(); ++oi) {
    CopyOp const& op = ops[oi];
    while (grp < group_end.size() && oi == group_end[grp]) {  // a lane's slice is complete: its lane may start
      MA_HIP(owner, hipStreamSynchronize(stream));
      if (stream2) MA_HIP(owner, hipStreamSynchronize(stream2));
      group_done(grp++);
    }
    if (!op.bytes || !op.src) continue;
    for (size_t o = 0; o < op.bytes; o += piece, ++queued) {
      hipStream_t const s_ = st[queued & 1];
      if (queued >= (stream2 ? 2u : 1u)) MA_HIP(owner, hipStreamSynchronize(s_));
      static int const gap_us = getenv("MA_UPLOAD_GAP_US") ? atoi(getenv("MA_UPLOAD_GAP_US")) : 0;
      if (gap_us > 0 && queued > 0) {  // (experiment: leave the link idle for a moment between pieces)
        auto const until = std::chrono::steady_clock::now() + std::chrono::microseconds(gap_us);
        while (std::chrono::steady_clock::now() < until) {}
      }
      MA_HIP(owner, hipMemcpyAsync(static_cast<char*>(op.dst) + o, static_cast<const char*>(op.src) + o, std::min(piece, op.bytes - o),
                                   hipMemcpyHostToDevice, s_));
    }
  }
  MA_HIP(owner, hipStreamSynchronize(stream));
  if (stream2) MA_HIP(owner, hipStreamSynchronize(stream2));
  while (grp < group_end.size()) group_done(grp++);
  if (getenv("MA_VERBOSE")) {
    size_t total = 0;
    for (CopyOp const& op : ops) total += op.src ? op.bytes : 0;
    double const ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    fprintf(stderr, "[microasm] t=%.1f upload done: %.1f MB in %zu pieces, %.2f ms (%.1f GB/s)\n", wall_ms(), total / 1e6, queued, ms,
            total / ms / 1e6);
  }
  return MA_OK;
}

// Plan the upload of the slice [w0, w1) of a host batch into input set `set` of lane `ch`: buffers reserved, the copies
// appended to `ops`; offsets stay absolute, the base pointers are moved back instead.  *d is the device view (valid once
// the copies have run).
int stage_lane_inputs(ma_ctx* ch, int set, const ma_batch_t* b, int w0, int w1, std::vector<CopyOp>* ops, DBatch* d) {
  int const n = w1 - w0;
  u32 const r0 = b->read_win_off[w0], r1 = b->read_win_off[w1];
  size_t const nr = static_cast<size_t>(r1) - r0;
  size_t const f0 = b->ref_off[w0], f1 = b->ref_off[w1];
  u64 const b0 = b->read_off[r0], b1 = b->read_off[r1];
  constexpr size_t kPad = 64;
  struct Item { const void* src; size_t bytes; bool padded; };
  InputSet& in = ch->in_sets[set];
  in.h_rwo.resize(static_cast<size_t>(n) + 1);
  for (int i = 0; i <= n; ++i) in.h_rwo[i] = b->read_win_off[w0 + i] - r0;
  Item const items[10] = {{b->ref_bases + f0, f1 - f0, true},
                          {b->ref_off + w0, 4ull * (n + 1), false},
                          {in.h_rwo.data(), 4ull * (n + 1), false},
                          {b->read_off + r0, 8ull * (nr + 1), false},
                          {b->read_bases + b0, static_cast<size_t>(b1 - b0), true},
                          {b->read_quals + b0, static_cast<size_t>(b1 - b0), true},
                          {b->read_qname_id + r0, 4 * nr, false},
                          {b->read_sample + r0, nr, false},
                          {b->read_flags + r0, nr, false},
                          {b->read_hint ? b->read_hint + r0 : nullptr, b->read_hint ? 4 * nr : 0, false}};
  char* dp[10] = {nullptr};
  for (int i = 0; i < 10; ++i) {
    if (i == 9 && !b->read_hint) break;
    size_t const front = items[i].padded ? kPad : 0;
    MA_HIP(ch, in.bufs[i].reserve(front + items[i].bytes + kPad + 16));
    char* base = static_cast<char*>(in.bufs[i].p);
    if (front) ops->push_back(CopyOp{base, nullptr, front});
    ops->push_back(CopyOp{base + front, items[i].src, items[i].bytes});
    if (items[i].padded) ops->push_back(CopyOp{base + front + items[i].bytes, nullptr, kPad});
    dp[i] = base + front;
  }
  *d = DBatch{};
  d->n_windows = n;
  d->n_reads = static_cast<i64>(nr);
  d->ref_bases = reinterpret_cast<const u8*>(dp[0]) - f0;
  d->ref_off = reinterpret_cast<const u32*>(dp[1]);
  d->read_win_off = reinterpret_cast<const u32*>(dp[2]);
  d->read_off = reinterpret_cast<const u64*>(dp[3]);
  d->read_bases = reinterpret_cast<const u8*>(dp[4]) - b0;
  d->read_quals = reinterpret_cast<const u8*>(dp[5]) - b0;
  d->read_qname_id = reinterpret_cast<const u32*>(dp[6]);
  d->read_sample = reinterpret_cast<const u8*>(dp[7]);
  d->read_flags = reinterpret_cast<const u8*>(dp[8]);
  d->read_hint = b->read_hint ? reinterpret_cast<const i32*>(dp[9]) : nullptr;
  return MA_OK;
}

// One batch on the host route: what every lane needs to compute its slice, and what it leaves for the delivery.
struct LaneResult {
  int rc = MA_OK;
  std::string err;
  int w0 = 0, w1 = 0;
  u32 r0 = 0;
  size_t nr = 0;
  u8* land = nullptr;                 // the lane's landing area of the job's set
  std::vector<size_t> dense_off;      // dense_table entry -> offset in land
  size_t aux_off = 0, packed_off = 0, packed_bytes = 0;
  std::vector<std::pair<const char*, float>> times;  // the lane's kernel timers of this job, resolved
  unsigned long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
struct UploadTask {  // the inputs of one prefetched batch, all lanes: carried out by the context's uploader thread
  std::vector<CopyOp> ops;           // lane after lane
  std::vector<size_t> lane_end;      // ops[.. lane_end[k]) are the slices of lanes 0 .. k
  std::mutex mu;
  std::condition_variable cv;
  size_t lanes_done = 0;             // the slices of lanes [0, lanes_done) are on the device
  bool done = false;
  int rc = MA_OK;
  std::string err;
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
  }
  void wait_lane(size_t k) {  // a lane starts as soon as ITS slice is there: the lanes of a batch start one after the other
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done || lanes_done > k; });
  }
};
struct HostJob {
  ma_batch_t batch{};                 // a copy of the caller's struct (its ARRAYS stay the caller's, valid until the call returns)
  const ma_batch_t* b = nullptr;
  int set = 0, lanes = 0;
  std::vector<int> wb;
  ma_params_t prm{};
  bool timing = false, accumulate = false, collect = false;
  double hbm_share = 1.0;
  u32 mask = 0;                       // seg_table entries to compute and pack
  std::shared_ptr<UploadTask> upload; // the upload ma_prefetch_batch started (null: every lane uploads its slice)
  std::vector<DBatch> staged;         // the lanes' device views of that upload
  const ma_geno_out_t* taps = nullptr;  // the caller's struct when it asked for the per-read taps (never on a job queued ahead)
  std::vector<LaneResult> res;
  std::mutex mu;
  std::condition_variable cv;
  int done = 0;
  void wait_all() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done == lanes; });
  }
};
struct LaneQueue {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::shared_ptr<HostJob>> q;
  bool stop = false;
  std::thread th;
};
struct HostAsync {
  std::vector<std::unique_ptr<LaneQueue>> lanes;
  std::shared_ptr<HostJob> jobs[2];   // by input set: queued (or computed) and not delivered yet
  std::shared_ptr<UploadTask> uploads[2];  // by input set: the upload of the batch the set holds
  std::mutex up_mu;
  std::condition_variable up_cv;
  std::deque<std::shared_ptr<UploadTask>> up_q;
  bool up_stop = false;
  std::thread uploader;
  u32 last_mask = 0;                  // what the last call asked for: a job queued ahead computes the same
  bool have_mask = false;
  std::vector<std::pair<const char*, float>> times;  // kernel timers of the delivered jobs (ma_last_kernel_times)
};
HostAsync* async_of(ma_ctx* ctx) {
  if (!ctx->host_async) ctx->host_async = new HostAsync();
  return static_cast<HostAsync*>(ctx->host_async);
}

// The lane's part of a job, on the lane's worker thread: inputs, the four stages, packed records into the landing area.
int lane_compute(ma_ctx* ch, HostJob& job, int k) {
  LaneResult& R = job.res[k];
  int const w0 = job.wb[k], w1 = job.wb[k + 1], set = job.set;
  const ma_batch_t* b = job.b;
  MA_HIP(ch, hipSetDevice(ch->device));
  ch->prm = job.prm;
  ch->timing = job.timing;
  ch->accumulate = job.accumulate;
  ch->collect = job.collect;
  ch->hbm_share = job.hbm_share;
  ch->timers_used = 0;
  for (auto& s : ch->stats) s = 0;
  ma_params_t const& p = ch->prm;
  int const n = w1 - w0;
  bool const verbose = getenv("MA_VERBOSE") != nullptr;
  auto const t_begin = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  double t_up = 0, t_gate = 0, t_asm = 0, t_msa = 0, t_geno = 0, t_pack = 0;
  u32 const r0 = b->read_win_off[w0], r1 = b->read_win_off[w1];
  size_t const nr = static_cast<size_t>(r1) - r0;
  size_t const f0 = b->ref_off[w0], f1 = b->ref_off[w1];
  u64 const b0 = b->read_off[r0], b1 = b->read_off[r1];
  R.w0 = w0; R.w1 = w1; R.r0 = r0; R.nr = nr;
  DBatch d{};
  if (job.upload) {
    job.upload->wait_lane(static_cast<size_t>(k));
    if (job.upload->rc != MA_OK) {
      ma_set_err(ch, "upload: " + job.upload->err);
      return job.upload->rc;
    }
    d = job.staged[k];
  } else {
    std::vector<CopyOp> ops;
    MA_TRY_RC(stage_lane_inputs(ch, set, b, w0, w1, &ops, &d));
    MA_TRY_RC(run_copy_ops(ch, ch->stream, nullptr, ops, std::vector<size_t>(), [](size_t) {}));
  }
  // ---- device-side outputs of the lane (fixed strides, as the kernels write them) ----
  LaneOut o;
  std::vector<OutField> const gf = gate_fields(p, n), af = asm_fields(p, n), vf = var_fields(p, n),
                              qf = geno_fields(p, n, static_cast<i64>(nr));
  std::vector<SegDesc> const segs = seg_table(p);
  MA_TRY_RC(alloc_fields(ch, &o.g, gf, 0, std::vector<bool>(gf.size(), true)));
  MA_TRY_RC(alloc_fields(ch, &o.a, af, 2, std::vector<bool>(af.size(), true)));
  MA_TRY_RC(alloc_fields(ch, &o.v, vf, 16, std::vector<bool>(vf.size(), true)));
  {
    std::vector<bool> need(qf.size(), false);
    need[0] = need[1] = true;  // allele_counts, var_qual; the taps and PL / GQ only if the caller asked for them
    if (job.taps) {
      need[2] = job.taps->aln_rec != nullptr; need[3] = job.taps->aln_cigar != nullptr;
      need[4] = job.taps->asg_allele != nullptr; need[5] = job.taps->asg_score != nullptr;
    }
    need[6] = (job.mask >> 17) & 1u;  // seg_table: var_pl, var_gq
    need[7] = (job.mask >> 18) & 1u;
    MA_TRY_RC(alloc_fields(ch, &o.q, qf, 32, need));
  }
  t_up = since();
  MA_TRY_RC(launch_gate(ch, d, o.g.max_approx, o.g.max_exact));
  t_gate = since();
  MA_TRY_RC(launch_assemble(ch, d, o.a, o.g.max_approx));
  t_asm = since();
  MA_TRY_RC(launch_msa(ch, d, o.a, o.v));
  t_msa = since();
  MA_TRY_RC(launch_genotype(ch, d, o.a, o.v, o.q));
  t_geno = since();
  // ---- results: the small dense arrays whole, the rest as packed records, all into the landing area ----
  OutPtrs const dev{{&o.g, &o.a, &o.v, &o.q}};
  size_t const N = n, MH = p.max_haps, MV = p.max_vars, MCG = p.max_cigar;
  std::vector<DenseDesc> const dense = dense_table(p);
  PackArgs D{};
  for (size_t i = 0; i < segs.size(); ++i) {
    void* dp = dev.at(segs[i].strct, segs[i].off);
    if (!((job.mask >> i) & 1u) || !dp) continue;
    D.seg[D.nseg++] = PackSeg{static_cast<const u8*>(dp), static_cast<u32>(segs[i].win_stride), static_cast<u32>(segs[i].unit),
                              segs[i].kind};
  }
  D.win_status = o.a.win_status; D.win_ncomp = o.a.win_ncomp; D.comp_hap0 = o.a.comp_hap0; D.comp_nhaps = o.a.comp_nhaps;
  D.hap_len = o.a.hap_len; D.hap_nruns = o.a.hap_nruns; D.win_nvars = o.v.win_nvars; D.var_ref_off = o.v.var_ref_off;
  D.var_ref_len = o.v.var_ref_len; D.var_nalts = o.v.var_nalts; D.alt_off = o.v.alt_off; D.alt_len = o.v.alt_len;
  D.MC = p.max_comps; D.MH = p.max_haps; D.MV = p.max_vars; D.MA = p.max_alts; D.MP = p.max_allele_bytes;
  MA_HIP(ch, ch->pack_aux.reserve(4 * (2 * N + 1) + 64));
  MA_TRY_RC(launch_pack_sizes(ch, D, n, ch->pack_aux.as<u32>()));
  // the landing area: [dense arrays][aux][records]; the records' size is only known on the device -- room for what the
  // last batch needed and half as much again (8 KB per window the first time), checked after the fact
  size_t const aux_bytes = 4 * (2 * N + 1);
  size_t land_bytes = 0;
  R.dense_off.assign(dense.size(), 0);
  for (size_t i = 0; i < dense.size(); ++i) {
    R.dense_off[i] = land_bytes;
    land_bytes += (dense[i].win_bytes * N + 15) & ~size_t(15);
  }
  R.aux_off = land_bytes;
  land_bytes += (aux_bytes + 15) & ~size_t(15);
  R.packed_off = land_bytes;
  size_t const guess = ch->last_packed ? ch->last_packed + ch->last_packed / 2 + 65536 : N * 8192 + 65536;
  MA_TRY_RC(ensure_pinned(ch, set, land_bytes + guess));
  R.land = static_cast<u8*>(ch->pin[set]);
  DenseCopies C{};
  for (size_t i = 0; i < dense.size(); ++i)
    C.c[C.n++] = DenseCopy{dev.at(dense[i].strct, dense[i].off), R.land + R.dense_off[i], static_cast<u32>(dense[i].win_bytes * N)};
  C.c[C.n++] = DenseCopy{ch->pack_aux.p, R.land + R.aux_off, static_cast<u32>(aux_bytes)};
  MA_TRY_RC(launch_pack_dense(ch, C));
  MA_TRY_RC(launch_pack_records(ch, D, n, ch->pack_aux.as<u32>(), R.land + R.packed_off, ch->pin_cap[set] - R.packed_off));
  t_pack = since();
  if (const ma_geno_out_t* uq = job.taps) {  // debug taps, straight into the caller's arrays (the call is waiting for them)
    auto d2h = [&](void* host, const void* devp, size_t bytes) -> int {
      if (host && devp && bytes) MA_HIP(ch, hipMemcpyAsync(host, devp, bytes, hipMemcpyDeviceToHost, ch->stream));
      return MA_OK;
    };
    MA_TRY_RC(d2h(uq->aln_rec ? uq->aln_rec + static_cast<size_t>(r0) * MH * 6 : nullptr, o.q.aln_rec, 4 * nr * MH * 6));
    MA_TRY_RC(d2h(uq->aln_cigar ? uq->aln_cigar + static_cast<size_t>(r0) * MH * (1 + MCG) : nullptr, o.q.aln_cigar,
                  4 * nr * MH * (1 + MCG)));
    MA_TRY_RC(d2h(uq->asg_allele ? uq->asg_allele + static_cast<size_t>(r0) * MV : nullptr, o.q.asg_allele, nr * MV));
    MA_TRY_RC(d2h(uq->asg_score ? uq->asg_score + static_cast<size_t>(r0) * MV : nullptr, o.q.asg_score, 8 * nr * MV));
  }
  MA_HIP(ch, ma_stream_sync(ch));
  size_t packed_bytes = static_cast<size_t>(reinterpret_cast<const u32*>(R.land + R.aux_off)[2 * N]) * 4u;
  if (packed_bytes > ch->pin_cap[set] - R.packed_off) {  // did not fit: a bigger landing area (its front part again), the records again
    MA_TRY_RC(ensure_pinned(ch, set, land_bytes + packed_bytes + 65536));
    R.land = static_cast<u8*>(ch->pin[set]);
    for (u32 i = 0; i < C.n; ++i) C.c[i].dst = R.land + (i < dense.size() ? R.dense_off[i] : R.aux_off);
    MA_TRY_RC(launch_pack_dense(ch, C));
    MA_TRY_RC(launch_pack_records(ch, D, n, ch->pack_aux.as<u32>(), R.land + R.packed_off, ch->pin_cap[set] - R.packed_off));
    MA_HIP(ch, ma_stream_sync(ch));
  }
  ch->last_packed = packed_bytes;
  R.packed_bytes = packed_bytes;
  // the lane's timers and counters belong to this job: the next job starts from zero while the caller reads these
  if (ch->timing) {
    R.times.reserve(ch->timers_used);
    for (size_t i = 0; i < ch->timers_used; ++i) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, ch->timers[i].beg, ch->timers[i].end);
      R.times.emplace_back(ch->timers[i].name, t);
    }
  }
  ch->timers_used = 0;
  for (int x = 0; x < 8; ++x) {
    R.stats[x] = ch->stats[x];
    ch->stats[x] = 0;
  }
  if (verbose)
    fprintf(stderr, "[microasm] t=%.1f host lane [%d, %d): enqueue-upload %.2f gate %.2f assemble %.2f msa %.2f genotype %.2f pack %.2f "
            "download %.2f ms; %.1f MB in, %.2f MB packed out\n", wall_ms(), w0, w1, t_up, t_gate - t_up, t_asm - t_gate,
            t_msa - t_asm, t_geno - t_msa, t_pack - t_geno, since() - t_pack,
            (static_cast<double>(f1 - f0) + 2.0 * static_cast<double>(b1 - b0) + 21.0 * nr) / 1e6, packed_bytes / 1e6);
  return MA_OK;
}

// The caller's part: the lane's landing area into the caller's arrays.
void lane_deliver(const ma_params_t& p, HostJob const& job, int k, OutPtrs const& user) {
  LaneResult const& R = job.res[k];
  size_t const N = static_cast<size_t>(R.w1 - R.w0), W0 = R.w0;
  if (N == 0) return;
  std::vector<DenseDesc> const dense = dense_table(p);
  for (size_t i = 0; i < dense.size(); ++i)
    if (void* hp = user.at(dense[i].strct, dense[i].off))
      std::memcpy(static_cast<u8*>(hp) + W0 * dense[i].win_bytes, R.land + R.dense_off[i], N * dense[i].win_bytes);
  std::vector<SegDesc> const segs = seg_table(p);
  PackArgs H{};
  for (size_t i = 0; i < segs.size(); ++i) {
    if (!((job.mask >> i) & 1u)) continue;
    u8* hp = static_cast<u8*>(user.at(segs[i].strct, segs[i].off));  // (non-null: the mask is the caller's)
    H.seg[H.nseg++] = PackSeg{hp + W0 * segs[i].win_stride, static_cast<u32>(segs[i].win_stride), static_cast<u32>(segs[i].unit),
                              segs[i].kind};
  }
  // what says how much of a window is in use: the dense arrays, from the landing area (the caller may have left some out)
  auto land_u32 = [&](int strct, size_t off) -> const u32* {
    for (size_t i = 0; i < dense.size(); ++i)
      if (dense[i].strct == strct && dense[i].off == off) return reinterpret_cast<const u32*>(R.land + R.dense_off[i]);
    return nullptr;
  };
  H.win_status = land_u32(1, off_of(&ma_asm_out_t::win_status));
  H.win_ncomp = land_u32(1, off_of(&ma_asm_out_t::win_ncomp));
  H.comp_hap0 = land_u32(1, off_of(&ma_asm_out_t::comp_hap0));
  H.comp_nhaps = land_u32(1, off_of(&ma_asm_out_t::comp_nhaps));
  H.hap_len = land_u32(1, off_of(&ma_asm_out_t::hap_len));
  H.hap_nruns = land_u32(1, off_of(&ma_asm_out_t::hap_nruns));
  H.win_nvars = land_u32(2, off_of(&ma_var_out_t::win_nvars));
  H.MC = p.max_comps; H.MH = p.max_haps; H.MV = p.max_vars; H.MA = p.max_alts; H.MP = p.max_allele_bytes;
  unpack_records(H, reinterpret_cast<const u32*>(R.land + R.aux_off), R.land + R.packed_off, static_cast<int>(N));
}

int ensure_lanes(ma_ctx* ctx, int lanes) {
  while (static_cast<int>(ctx->lanes.size()) < lanes) {
    ma_ctx* ch = new (std::nothrow) ma_ctx();
    if (!ch) return MA_ERR_NOMEM;
    ch->device = ctx->device;
    ch->memspace = MA_MEM_DEVICE;
    MA_HIP(ctx, hipStreamCreateWithFlags(&ch->stream, hipStreamNonBlocking));
    MA_HIP(ctx, hipEventCreateWithFlags(&ch->lane_done, hipEventDisableTiming));
    ctx->lanes.push_back(ch);
  }
  return MA_OK;
}

void lane_worker(ma_ctx* ch, LaneQueue* lq, int k) {
  while (true) {
    std::shared_ptr<HostJob> job;
    {
      std::unique_lock<std::mutex> lk(lq->mu);
      lq->cv.wait(lk, [&] { return lq->stop || !lq->q.empty(); });
      if (lq->q.empty()) return;  // stop, nothing left
      job = lq->q.front();
      lq->q.pop_front();
    }
    LaneResult& R = job->res[k];
    R.rc = job->wb[k + 1] > job->wb[k] ? lane_compute(ch, *job, k) : MA_OK;
    if (R.rc != MA_OK) {
      R.err = ma_get_err(ch);
      (void)hipStreamSynchronize(ch->stream);  // whatever was queued before the error is done before the buffers are reused
    }
    {
      std::lock_guard<std::mutex> lk(job->mu);
      job->done++;
    }
    job->cv.notify_all();
  }
}

int ensure_workers(ma_ctx* ctx, int lanes) {
  MA_TRY_RC(ensure_lanes(ctx, lanes));
  HostAsync* ha = async_of(ctx);
  while (static_cast<int>(ha->lanes.size()) < lanes) {
    int const k = static_cast<int>(ha->lanes.size());
    ha->lanes.emplace_back(new LaneQueue());
    LaneQueue* lq = ha->lanes.back().get();
    lq->th = std::thread(lane_worker, ctx->lanes[k], lq, k);
  }
  return MA_OK;
}

void uploader_loop(ma_ctx* ctx, HostAsync* ha) {
  (void)hipSetDevice(ctx->device);
  while (true) {
    std::shared_ptr<UploadTask> t;
    {
      std::unique_lock<std::mutex> lk(ha->up_mu);
      ha->up_cv.wait(lk, [&] { return ha->up_stop || !ha->up_q.empty(); });
      if (ha->up_q.empty()) return;
      t = ha->up_q.front();
      ha->up_q.pop_front();
    }
    int rc = MA_OK;
    {
      auto run = [&]() -> int {
        MA_HIP(ctx, hipSetDevice(ctx->device));
        return run_copy_ops(ctx, ctx->copy_stream, getenv("MA_UPLOAD_TWO_STREAMS") ? ctx->copy_stream2 : nullptr, t->ops, t->lane_end,
                            [&](size_t g) {
                              {
                                std::lock_guard<std::mutex> lk(t->mu);
                                t->lanes_done = g + 1;
                              }
                              t->cv.notify_all();
                            });
      };
      rc = run();
    }
    {
      std::lock_guard<std::mutex> lk(t->mu);
      t->rc = rc;
      if (rc != MA_OK) t->err = ma_get_err(ctx);
      t->done = true;
    }
    t->cv.notify_all();
  }
}

// wait for every job that is queued or running (their results stay with the job until a call asks for them or drops them)
void drain_host(ma_ctx* ctx) {
  if (!ctx->host_async) return;
  HostAsync* ha = static_cast<HostAsync*>(ctx->host_async);
  for (auto& j : ha->jobs)
    if (j) j->wait_all();
  for (auto& u : ha->uploads)
    if (u) u->wait();
}

void stop_workers(ma_ctx* ctx) {
  if (!ctx->host_async) return;
  HostAsync* ha = static_cast<HostAsync*>(ctx->host_async);
  for (auto& lq : ha->lanes) {
    {
      std::lock_guard<std::mutex> lk(lq->mu);
      lq->stop = true;
    }
    lq->cv.notify_all();
    if (lq->th.joinable()) lq->th.join();
  }
  {
    std::lock_guard<std::mutex> lk(ha->up_mu);
    ha->up_stop = true;
  }
  ha->up_cv.notify_all();
  if (ha->uploader.joinable()) ha->uploader.join();
  delete ha;
  ctx->host_async = nullptr;
}

int host_lanes(const ma_ctx* ctx, int n_windows) {
  // the caller's stream carries nothing on this route: the lanes (and the prefetch stream) have the hardware queues to
  // themselves -- four by default (GPU_MAX_HW_QUEUES), so three lanes + the copy stream unless the host raised it
  static int const hwq = getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
  int lanes = ctx->n_lanes > 0 ? ctx->n_lanes : (n_windows >= 2048 ? (hwq >= 6 ? 4 : 3) : (n_windows >= 512 ? 2 : 1));
  if (const char* e = getenv("MA_STREAMS")) lanes = atoi(e) > 0 ? atoi(e) : lanes;
  if (lanes > n_windows / 2) lanes = n_windows / 2 > 0 ? n_windows / 2 : 1;
  return lanes > 8 ? 8 : lanes;
}

std::vector<int> lane_bounds(int n_windows, int lanes) {
  std::vector<int> wb(lanes + 1);
  for (int k = 0; k <= lanes; ++k) wb[k] = static_cast<int>(static_cast<long long>(n_windows) * k / lanes);
  return wb;
}

// staged device views of a prefetched batch, per set and lane (parent context; kept beside pf_batch)
struct PrefetchViews {
  std::vector<DBatch> d[2];
};
std::mutex g_views_mu;
std::unordered_map<ma_ctx*, PrefetchViews> g_views;
PrefetchViews& views_of(ma_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_views_mu);
  return g_views[ctx];
}
void forget_views(ma_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_views_mu);
  g_views.erase(ctx);
}

// queue the lanes' jobs of batch `b`, whose inputs are (being) staged in `set` when `ready`
std::shared_ptr<HostJob> submit_host(ma_ctx* ctx, int lanes, const ma_batch_t* b, int set, std::shared_ptr<UploadTask> upload,
                                     u32 mask, const ma_geno_out_t* taps) {
  HostAsync* ha = async_of(ctx);
  auto job = std::make_shared<HostJob>();
  job->batch = *b;
  job->b = &job->batch;
  job->set = set;
  job->lanes = lanes;
  job->wb = lane_bounds(b->n_windows, lanes);
  job->prm = ctx->prm;
  job->timing = ctx->timing;
  job->accumulate = ctx->accumulate;
  job->collect = ctx->collect;
  job->hbm_share = ctx->hbm_share / lanes;
  job->mask = mask;
  job->upload = upload;
  if (upload) job->staged = views_of(ctx).d[set];
  job->taps = taps;
  job->res.resize(lanes);
  for (int k = 0; k < lanes; ++k) {
    LaneQueue* lq = ha->lanes[k].get();
    {
      std::lock_guard<std::mutex> lk(lq->mu);
      lq->q.push_back(job);
    }
    lq->cv.notify_one();
  }
  return job;
}

int process_host(ma_ctx* ctx, int lanes, const ma_batch_t* b, const ma_gate_out_t* g, const ma_asm_out_t* a,
                 const ma_var_out_t* v, const ma_geno_out_t* q) {
  MA_TRY_RC(ensure_workers(ctx, lanes));
  HostAsync* ha = async_of(ctx);
  OutPtrs const user{{g, a, v, q}};
  u32 const mask = seg_mask_of(ctx->prm, user);
  bool const taps = wants_taps(q);
  // did ma_prefetch_batch upload this batch?  (the oldest set that holds it); otherwise any set that holds nothing
  int set = -1;
  bool prefetched = false;
  for (int s = 0; s < 2; ++s)
    if (ctx->pf_batch[s] == b && ctx->pf_sig[s][0] == b->n_windows && ctx->pf_sig[s][1] == b->n_reads && ctx->pf_sig[s][2] == lanes &&
        (set < 0 || ctx->pf_seq[s] < ctx->pf_seq[set])) {
      set = s;
      prefetched = true;
    }
  if (set < 0) {
    set = !ctx->pf_batch[0] ? 0 : (!ctx->pf_batch[1] ? 1 : (ctx->pf_seq[0] < ctx->pf_seq[1] ? 0 : 1));
    if (ctx->pf_batch[set]) {  // both sets hold batches that were never processed: give the older one up
      if (ha->jobs[set]) ha->jobs[set]->wait_all();
      ha->jobs[set].reset();
      if (ha->uploads[set]) ha->uploads[set]->wait();
      ha->uploads[set].reset();
      ctx->pf_batch[set] = nullptr;
    }
  }
  std::shared_ptr<HostJob> job = ha->jobs[set];
  if (job && (job->mask != mask || taps || std::memcmp(&job->prm, &ctx->prm, sizeof(ma_params_t)) != 0 ||
              job->timing != ctx->timing || job->accumulate != ctx->accumulate || job->collect != ctx->collect)) {
    job->wait_all();  // queued ahead under other assumptions than this call's: computed again below, from the staged inputs
    job.reset();
  }
  if (!prefetched) ha->uploads[set].reset();
  if (!job) job = submit_host(ctx, lanes, b, set, prefetched ? ha->uploads[set] : nullptr, mask, taps ? q : nullptr);
  ha->jobs[set] = job;
  ha->last_mask = mask;
  ha->have_mask = true;
  job->wait_all();
  if (!ctx->accumulate) ha->times.clear();
  int rc = MA_OK;
  for (int k = 0; k < lanes; ++k) {
    LaneResult const& R = job->res[k];
    if (R.rc != MA_OK && rc == MA_OK) {
      ma_set_err(ctx, "lane " + std::to_string(k) + ": " + R.err);
      rc = R.rc;
    }
    ha->times.insert(ha->times.end(), R.times.begin(), R.times.end());
    for (int x = 0; x < 8; ++x) ctx->stats[x] += R.stats[x];
  }
  if (rc == MA_OK) {
    bool const verbose = getenv("MA_VERBOSE") != nullptr;
    auto const t0 = std::chrono::steady_clock::now();
    if (lanes > 1) {  // (one scatter thread per lane: ~0.7 ms each, side by side)
      std::vector<std::thread> th;
      for (int k = 0; k < lanes; ++k) th.emplace_back([&, k]() { lane_deliver(ctx->prm, *job, k, user); });
      for (auto& t : th) t.join();
    } else {
      lane_deliver(ctx->prm, *job, 0, user);
    }
    if (verbose)
      fprintf(stderr, "[microasm] t=%.1f host route: records of %d windows scattered in %.2f ms\n", wall_ms(), b->n_windows,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  ha->jobs[set].reset();
  ha->uploads[set].reset();
  ctx->pf_batch[set] = nullptr;  // consumed (or used as plain staging): free for the next prefetch
  return rc;
}

}  // namespace

extern "C" {

void ma_default_params(ma_params_t* p) {
  std::memset(p, 0, sizeof(*p));
  p->min_k = 13; p->max_k = 127; p->k_step = 6;        // graph_params.h:11-26
  p->min_node_cov = 2; p->min_anchor_cov = 5;          // graph_params.h:17-20
  p->num_samples = 2;
  p->min_anchor_len = 150;                              // graph.cpp:88
  p->max_mismatch = 2;                                  // graph.h:129
  p->bfs_limit = 1 << 20;                               // max_flow.h:69
  p->aln_tier = 0; p->min_aln_score = 80;
  p->max_comps = 4; p->max_haps = 16; p->max_hap_len = 2048; p->max_runs = 256;
  p->max_vars = 64; p->max_alts = 4; p->max_allele_bytes = 4096; p->max_cigar = 16;
  p->case_ctrl_mode = 1;
}

int ma_create(const ma_params_t* prm, int device, int memspace, ma_ctx_t** out) {
  if (!prm || !out) return MA_ERR_ARG;
  *out = nullptr;
  int rc = check_params(*prm);
  if (rc != MA_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    return MA_ERR_NO_DEVICE;  // the engine has no CPU fallback, by design
  }
  if (hipSetDevice(device) != hipSuccess) return MA_ERR_HIP;
  ma_ctx* c = new (std::nothrow) ma_ctx();
  if (!c) return MA_ERR_NOMEM;
  c->prm = *prm;
  c->device = device;
  c->memspace = memspace;
  // own non-blocking stream: the legacy null stream would serialise this context against every other context and
  // copy of the process (two feeders on one device would never overlap); ma_set_stream replaces it
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;  // (falling back to the null stream silently would change how the context orders against the caller's work)
    return MA_ERR_HIP;
  }
  c->stream = c->own_stream;
  *out = c;
  return MA_OK;
}

void ma_destroy(ma_ctx_t* ctx) {
  if (!ctx) return;
  stop_workers(ctx);  // (the lanes' threads finish what is queued, then end)
  delete static_cast<DevLanePool*>(ctx->dev_pool);
  ctx->dev_pool = nullptr;
  forget_views(ctx);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& b : ctx->in_stage) b.release();
  for (auto& b : ctx->out_stage) b.release();
  ctx->ws_build.release(); ctx->ws_nodes.release(); ctx->ws_clean.release();
  ctx->spec_data.release(); ctx->spec_out.release(); ctx->spec_nodes.release();
  ctx->ws_aln.release(); ctx->ws_misc.release(); ctx->ws_cx.release(); ctx->ws_gen.release(); ctx->ws_mm.release(); ctx->dev_stats.release();
  ctx->pack_aux.release();
  for (auto& pp : ctx->pin) {
    if (pp) (void)hipHostFree(pp);
    pp = nullptr;
  }
  for (hipStream_t cs : {ctx->copy_stream, ctx->copy_stream2})
    if (cs) {
      (void)hipStreamSynchronize(cs);
      (void)hipStreamDestroy(cs);
    }
  for (auto& st : ctx->in_sets)
    for (auto& bf : st.bufs) bf.release();
  for (auto& t : ctx->timers) {
    (void)hipEventDestroy(t.beg);
    (void)hipEventDestroy(t.end);
  }
  for (ma_ctx* ch : ctx->lanes) {
    hipStream_t const cs = ch->stream;
    ch->lane_rwo.release();
    ma_destroy(ch);
    if (cs) (void)hipStreamDestroy(cs);
  }
  if (ctx->lane_done) (void)hipEventDestroy(ctx->lane_done);
  if (ctx->hi_ev) (void)hipEventDestroy(ctx->hi_ev);
  if (ctx->hi_stream) (void)hipStreamDestroy(ctx->hi_stream);
  if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char* ma_last_error(const ma_ctx_t* ctx) {
  if (!ctx) return "null context";
  // a copy per calling thread: the context's string may be rewritten by a worker thread while the caller reads it
  static thread_local std::string copy;
  copy = ma_get_err(const_cast<ma_ctx_t*>(ctx));
  return copy.c_str();
}

int ma_set_stream(ma_ctx_t* ctx, void* s) {
  if (!ctx) return MA_ERR_ARG;
  ctx->stream = static_cast<hipStream_t>(s);
  // the context's own stream is not needed any more; an idle stream still takes one of the process's few hardware queues
  // (4 by default) away from the lanes
  if (ctx->own_stream && ctx->stream != ctx->own_stream) {
    (void)hipStreamSynchronize(ctx->own_stream);
    (void)hipStreamDestroy(ctx->own_stream);
    ctx->own_stream = nullptr;
  }
  return MA_OK;
}

int ma_synchronize(ma_ctx_t* ctx) {
  if (!ctx) return MA_ERR_ARG;
  MA_HIP(ctx, ma_stream_sync(ctx));
  return MA_OK;
}

int ma_timing_control(ma_ctx_t* ctx, int mode) {
  if (!ctx) return MA_ERR_ARG;
  drain_host(ctx);  // (the lanes' timers and counters are theirs while a job runs)
  if (ctx->host_async) static_cast<HostAsync*>(ctx->host_async)->times.clear();
  ctx->timing = mode != 0;
  ctx->accumulate = mode >= 2;
  ctx->collect = mode == 3;
  ctx->dev_stats_clean = false;
  ctx->timers_used = 0;
  for (auto& v : ctx->stats) v = 0;
  for (ma_ctx* ch : ctx->lanes) ma_timing_control(ch, mode);
  return MA_OK;
}

int ma_set_streams(ma_ctx_t* ctx, int n) {
  if (!ctx || n < 0 || n > 8) return MA_ERR_ARG;
  ctx->n_lanes = n;
  return MA_OK;
}

int ma_last_stats(ma_ctx_t* ctx, unsigned long long* out, int cap) {
  if (!ctx || !out) return MA_ERR_ARG;
  drain_host(ctx);
  int n = 0;
  for (; n < cap && n < 8; ++n) {
    out[n] = ctx->stats[n];
    for (ma_ctx* ch : ctx->lanes) out[n] += ch->stats[n];
  }
  // entries 8..13: device-side workload statistics (mode 3 only; zero otherwise)
  auto add_dev = [&](ma_ctx* c) -> int {
    if (!c->dev_stats.p || !c->dev_stats_clean) return MA_OK;
    unsigned long long h[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    MA_HIP(c, hipSetDevice(c->device));
    MA_HIP(c, ma_stream_sync(c));
    MA_HIP(c, hipMemcpy(h, c->dev_stats.p, sizeof(h), hipMemcpyDeviceToHost));
    for (int x = 0; x < 12 && 8 + x < cap; ++x) out[8 + x] += h[x];
    return MA_OK;
  };
  if (cap > 8) {
    for (int x = 8; x < cap && x < 20; ++x) out[x] = 0;
    if (add_dev(ctx) != MA_OK) return MA_ERR_HIP;
    for (ma_ctx* ch : ctx->lanes)
      if (add_dev(ch) != MA_OK) return MA_ERR_HIP;
    n = cap < 20 ? cap : 20;
  }
  return n;
}

int ma_last_kernel_times(ma_ctx_t* ctx, const char** names, float* ms, int cap) {
  if (!ctx) return MA_ERR_ARG;
  MA_HIP(ctx, ma_stream_sync(ctx));
  int n = 0;
  for (size_t i = 0; i < ctx->timers_used && n < cap; ++i, ++n) {
    names[n] = ctx->timers[i].name;
    float t = 0.f;
    (void)hipEventElapsedTime(&t, ctx->timers[i].beg, ctx->timers[i].end);
    ms[n] = t;
  }
  if (ctx->host_async) {  // host route: the lanes' timers were resolved into the jobs that ma_process_batch delivered
    for (auto const& t : static_cast<HostAsync*>(ctx->host_async)->times) {
      if (n >= cap) break;
      names[n] = t.first;
      ms[n++] = t.second;
    }
    return n;
  }
  for (ma_ctx* ch : ctx->lanes) {  // kernels launched on the child lanes of ma_process_batch
    for (size_t i = 0; i < ch->timers_used && n < cap; ++i, ++n) {
      names[n] = ch->timers[i].name;
      float t = 0.f;
      (void)hipEventElapsedTime(&t, ch->timers[i].beg, ch->timers[i].end);
      ms[n] = t;
    }
  }
  return n;
}

#define MA_BEGIN(ctx)                                   \
  if (!(ctx)) return MA_ERR_ARG;                        \
  MA_HIP(ctx, hipSetDevice((ctx)->device));             \
  if (!(ctx)->accumulate) {                             \
    (ctx)->timers_used = 0;                             \
    if (!(ctx)->host_async)                             \
      for (ma_ctx* _ch : (ctx)->lanes) _ch->timers_used = 0; \
  }

#define MA_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc != MA_OK) return _rc; \
  } while (0)

int ma_repeat_gate_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_gate_out_t* out) {
  MA_BEGIN(ctx);
  if (!out || !out->max_approx || !out->max_exact) return MA_ERR_ARG;
  DBatch d;
  MA_TRY(stage_batch(ctx, b, &d));
  OutMirror<ma_gate_out_t> g;
  MA_TRY(g.prepare(ctx, out, gate_fields(ctx->prm, d.n_windows), 0, false));
  MA_TRY(launch_gate(ctx, d, g.dev.max_approx, g.dev.max_exact));
  MA_TRY(g.download(ctx));
  if (ctx->memspace == MA_MEM_HOST) MA_HIP(ctx, ma_stream_sync(ctx));
  return MA_OK;
}

int ma_assemble_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_asm_out_t* out) {
  MA_BEGIN(ctx);
  if (!out) return MA_ERR_ARG;
  DBatch d;
  MA_TRY(stage_batch(ctx, b, &d));
  OutMirror<ma_asm_out_t> a;
  MA_TRY(a.prepare(ctx, out, asm_fields(ctx->prm, d.n_windows), 2, false));
  // the k-cascade consumes the gate result; compute it into scratch
  MA_HIP(ctx, ctx->ws_misc.reserve(8ull * (d.n_windows + 1)));
  u32* approx = ctx->ws_misc.as<u32>();
  u32* exact = approx + d.n_windows;
  MA_TRY(launch_gate(ctx, d, approx, exact));
  MA_TRY(launch_assemble(ctx, d, a.dev, approx));
  MA_TRY(a.download(ctx));
  if (ctx->memspace == MA_MEM_HOST) MA_HIP(ctx, ma_stream_sync(ctx));
  return MA_OK;
}

int ma_msa_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_asm_out_t* asmb, const ma_var_out_t* out) {
  MA_BEGIN(ctx);
  if (!out || !asmb) return MA_ERR_ARG;
  DBatch d;
  MA_TRY(stage_batch(ctx, b, &d));
  OutMirror<ma_asm_out_t> a;
  MA_TRY(a.prepare(ctx, asmb, asm_fields(ctx->prm, d.n_windows), 2, true));
  OutMirror<ma_var_out_t> v;
  MA_TRY(v.prepare(ctx, out, var_fields(ctx->prm, d.n_windows), 16, false));
  MA_TRY(launch_msa(ctx, d, a.dev, v.dev));
  MA_TRY(v.download(ctx));
  // win_status may gain MA_W_VAR_OVERFLOW
  if (ctx->memspace == MA_MEM_HOST) {
    MA_HIP(ctx, hipMemcpyAsync(asmb->win_status, a.dev.win_status, 4ull * d.n_windows, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, ma_stream_sync(ctx));
  }
  return MA_OK;
}

int ma_genotype_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_asm_out_t* asmb, const ma_var_out_t* vars,
                      const ma_geno_out_t* out) {
  MA_BEGIN(ctx);
  if (!out || !asmb || !vars) return MA_ERR_ARG;
  DBatch d;
  MA_TRY(stage_batch(ctx, b, &d));
  OutMirror<ma_asm_out_t> a;
  MA_TRY(a.prepare(ctx, asmb, asm_fields(ctx->prm, d.n_windows), 2, true));
  OutMirror<ma_var_out_t> v;
  MA_TRY(v.prepare(ctx, vars, var_fields(ctx->prm, d.n_windows), 16, true));
  OutMirror<ma_geno_out_t> g;
  MA_TRY(g.prepare(ctx, out, geno_fields(ctx->prm, d.n_windows, d.n_reads), 32, false));
  MA_TRY(launch_genotype(ctx, d, a.dev, v.dev, g.dev));
  MA_TRY(g.download(ctx));
  // win_status may gain MA_W_CIGAR_OVERFLOW / MA_W_READ_OVERFLOW
  if (ctx->memspace == MA_MEM_HOST) {
    MA_HIP(ctx, hipMemcpyAsync(asmb->win_status, a.dev.win_status, 4ull * d.n_windows, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(ctx, ma_stream_sync(ctx));
  }
  return MA_OK;
}

int ma_annotate_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_asm_out_t* asmb, const ma_var_out_t* vars,
                      double gc_frac, const ma_cx_out_t* out) {
  MA_BEGIN(ctx);
  if (!out || !asmb || !vars || !b || b->n_windows < 0) return MA_ERR_ARG;
  if (!out->seq_cx_i || !out->seq_cx_f || !out->seq_cx_d || !out->graph_cx) return MA_ERR_ARG;
  if (!(gc_frac == gc_frac)) return MA_ERR_ARG;  // NaN
  DBatch d{};
  d.n_windows = b->n_windows;  // only the window count is read: the inputs are the assembly + variant outputs
  OutMirror<ma_asm_out_t> a;
  MA_TRY(a.prepare(ctx, asmb, asm_fields(ctx->prm, d.n_windows), 2, true));
  OutMirror<ma_var_out_t> v;
  MA_TRY(v.prepare(ctx, vars, var_fields(ctx->prm, d.n_windows), 16, true));
  OutMirror<ma_cx_out_t> c;
  MA_TRY(c.prepare(ctx, out, cx_fields(ctx->prm, d.n_windows), 40, false));
  MA_TRY(launch_annotate(ctx, d, a.dev, v.dev, gc_frac, c.dev));
  MA_TRY(c.download(ctx));
  if (ctx->memspace == MA_MEM_HOST) MA_HIP(ctx, ma_stream_sync(ctx));
  return MA_OK;
}

int ma_prefetch_batch(ma_ctx_t* ctx, const ma_batch_t* next) {
  if (!ctx || !next) return MA_ERR_ARG;
  if (ctx->memspace != MA_MEM_HOST || next->n_windows <= 0 || getenv("MA_HOST_LEGACY")) return MA_OK;  // nothing to stage
  MA_HIP(ctx, hipSetDevice(ctx->device));
  int const set = !ctx->pf_batch[0] ? 0 : (!ctx->pf_batch[1] ? 1 : -1);
  if (set < 0) return MA_OK;  // two batches are waiting already
  if (!ctx->copy_stream) MA_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (!ctx->copy_stream2) MA_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream2, hipStreamNonBlocking));
  int const lanes = host_lanes(ctx, next->n_windows);
  MA_TRY_RC(ensure_workers(ctx, lanes));
  HostAsync* ha = async_of(ctx);
  std::vector<int> const wb = lane_bounds(next->n_windows, lanes);
  PrefetchViews& pv = views_of(ctx);
  pv.d[set].assign(lanes, DBatch{});
  auto task = std::make_shared<UploadTask>();
  for (int k = 0; k < lanes; ++k) {
    if (wb[k + 1] <= wb[k]) {
      task->lane_end.push_back(task->ops.size());
      continue;
    }
    int const rc = stage_lane_inputs(ctx->lanes[k], set, next, wb[k], wb[k + 1], &task->ops, &pv.d[set][k]);
    if (rc != MA_OK) {
      ma_set_err(ctx, "prefetch, lane " + std::to_string(k) + ": " + ma_get_err(ctx->lanes[k]));
      return rc;
    }
    task->lane_end.push_back(task->ops.size());
  }
  // the copies themselves are the uploader thread's: piece by piece, so that the copy engine's queue stays short
  if (!ha->uploader.joinable()) ha->uploader = std::thread(uploader_loop, ctx, ha);
  {
    std::lock_guard<std::mutex> lk(ha->up_mu);
    ha->up_q.push_back(task);
  }
  ha->up_cv.notify_one();
  ha->uploads[set] = task;
  ctx->pf_batch[set] = next;
  ctx->pf_sig[set][0] = next->n_windows;
  ctx->pf_sig[set][1] = next->n_reads;
  ctx->pf_sig[set][2] = lanes;
  ctx->pf_seq[set] = ++ctx->pf_counter;
  // ... and its compute jobs behind whatever the lanes are doing: the call that brings the batch will find the records
  // waiting (or on their way).  Not in the statistics-gathering mode of the bench (its device counters are per call).
  ha->jobs[set].reset();
  if (ha->have_mask && !ctx->collect && !getenv("MA_NO_RUNAHEAD"))
    ha->jobs[set] = submit_host(ctx, lanes, next, set, task, ha->last_mask, nullptr);
  return MA_OK;
}

int ma_process_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_gate_out_t* gate, const ma_asm_out_t* asmb,
                     const ma_var_out_t* vars, const ma_geno_out_t* geno) {
  MA_BEGIN(ctx);
  if (!gate || !asmb || !vars || !geno) return MA_ERR_ARG;
  if (!gate->max_approx || !gate->max_exact || !geno->allele_counts || !geno->var_qual) return MA_ERR_ARG;
  if (ctx->memspace == MA_MEM_HOST && !getenv("MA_HOST_LEGACY")) {
    // the host route: every lane uploads its own slice, computes, and brings back packed records (see process_host).
    // The caller's stream carries nothing here, so all four default hardware queues are the lanes'.
    if (!b || b->n_windows < 0) return MA_ERR_ARG;
    if (b->n_windows == 0) return MA_OK;
    if (!asmb->win_status || !asmb->win_ncomp || !asmb->comp_hap0 || !asmb->comp_nhaps || !asmb->hap_len ||
        !asmb->hap_nruns || !vars->win_nvars)
      return MA_ERR_ARG;
    return process_host(ctx, host_lanes(ctx, b->n_windows), b, gate, asmb, vars, geno);
  }
  DBatch d;
  MA_TRY(stage_batch(ctx, b, &d));
  OutMirror<ma_gate_out_t> g;
  MA_TRY(g.prepare(ctx, gate, gate_fields(ctx->prm, d.n_windows), 0, false));
  OutMirror<ma_asm_out_t> a;
  MA_TRY(a.prepare(ctx, asmb, asm_fields(ctx->prm, d.n_windows), 2, false));
  OutMirror<ma_var_out_t> v;
  MA_TRY(v.prepare(ctx, vars, var_fields(ctx->prm, d.n_windows), 16, false));
  OutMirror<ma_geno_out_t> q;
  MA_TRY(q.prepare(ctx, geno, geno_fields(ctx->prm, d.n_windows, d.n_reads), 32, false));
  // Automatic: three lanes for big batches -- the process has four hardware queues by default (ROCm's GPU_MAX_HW_QUEUES) and a
  // fourth lane would share one with the caller's stream (measured: -18 %); a host that raises GPU_MAX_HW_QUEUES to >= 6
  // before HIP starts gets four (+2 %).
  // (read once: the variable only means something if it was set before HIP started)
  static int const hwq = getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
  // Round 6, since a lane's windows go through every stage in ONE chunk: with a single k (no ladder) TWO lanes of 8192 windows
  // beat four of 4096 (274 against 268 k windows/s on the headline workload; 2 / 3 / 4 / 5 / 6 lanes: 274 / 269 / 268 / 257 /
  // 256) -- every launch has a tail and a latency floor that twice the windows amortise, and two lanes still overlap one
  // stage's tail with another's body.  The ladder's many small rungs want the four (145 against 140 k).
  bool const single_k = ctx->prm.min_k == ctx->prm.max_k;
  int lanes = ctx->n_lanes > 0 ? ctx->n_lanes
                               : (d.n_windows >= 8192 && hwq >= 6 ? (single_k ? 2 : 4) : (d.n_windows >= 6144 ? 3 : (d.n_windows >= 2048 ? 2 : 1)));
  if (const char* e = getenv("MA_STREAMS")) lanes = atoi(e) > 0 ? atoi(e) : lanes;
  if (lanes > d.n_windows / 2) lanes = d.n_windows / 2 > 0 ? d.n_windows / 2 : 1;
  if (lanes > 8) lanes = 8;
  if (lanes <= 1) {
    MA_TRY(launch_gate(ctx, d, g.dev.max_approx, g.dev.max_exact));
    MA_TRY(launch_assemble(ctx, d, a.dev, g.dev.max_approx));
    MA_TRY(launch_msa(ctx, d, a.dev, v.dev));
    MA_TRY(launch_genotype(ctx, d, a.dev, v.dev, q.dev));
  } else {
    MA_TRY(process_in_lanes(ctx, lanes, d, g.dev, a.dev, v.dev, q.dev));
  }
  MA_TRY(g.download(ctx));
  MA_TRY(a.download(ctx));
  MA_TRY(v.download(ctx));
  MA_TRY(q.download(ctx));
  if (ctx->memspace == MA_MEM_HOST) MA_HIP(ctx, ma_stream_sync(ctx));
  return MA_OK;
}

}  // extern "C"
