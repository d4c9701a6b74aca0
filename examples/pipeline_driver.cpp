// pipeline_driver.cpp -- the `pipeline` host around the engine: reference + alignment files in,
// coordinate-sorted de-duplicated variant records out.  Three stages run side by side, as the reference's AsyncWorker /
// PipelineExecutor do at window granularity (core/async_worker.cpp:47-110, core/pipeline_executor.cpp:174-252):
//   extract   WindowBuilder tiles the regions; per window the skip gates (N-only, max-k repeat, inactive region), the read
//             collector (filters, coverage-capped paired downsampling, comparator) and the coverage gate; the windows that
//             are left are flattened into batches
//   engine    ma_prefetch_batch(next) + ma_process_batch(this): the next batch uploads under this batch's kernels
//   flush     results -> records -> VariantStore (same CHROM+POS+REF from overlapping windows: the better covered call wins)
//             -> everything before the next batch's first window is written, in coordinate order
// Build (no HIP headers needed; zlib only for BAM input):
//   g++ -std=c++17 -O2 examples/pipeline_driver.cpp -Iinclude -Llancet2_amd -lmicroasm -Wl,-rpath,$PWD/lancet2_amd
//       -Wl,--allow-shlib-undefined -DLANCET2_AMD_WITH_ZLIB -lz -lpthread -o pipeline_driver
//   ./pipeline_driver --reference ref.fa --normal n.sam --tumor t.sam --region chr1:1-20000 --out calls.tsv
// Flags follow the reference CLI (cli/cli_interface.cpp:203-303) where the engine has the knob.
#include <malloc.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "../lancet2_amd/host/pipeline_host.hpp"

using namespace lancet2_amd::host;

namespace {

struct Outputs {  // caller-owned fixed-stride result buffers of one batch
  std::vector<uint32_t> u32;
  std::vector<double> f64;
  std::vector<uint8_t> u8;
  std::vector<int32_t> i32;
  ma_gate_out_t gate{};
  ma_asm_out_t asmb{};
  ma_var_out_t vars{};
  ma_geno_out_t geno{};
  ma_cx_out_t cx{};
  std::vector<int32_t> cx_i;
  std::vector<float> cx_f;
  std::vector<double> cx_d;
  void Allocate(const ma_params_t& p, int n) {
    size_t const N = n, MC = p.max_comps, MH = p.max_haps, ML = p.max_hap_len, MR = p.max_runs, MV = p.max_vars, MA = p.max_alts,
                 MP = p.max_allele_bytes, S = p.num_samples;
    size_t const G = (MA + 1) * (MA + 2) / 2;
    u32.assign(2 * N + 3 * N + 6 * N * MC + 2 * N * MH + 2 * N * MH * MR + N + 6 * N * MV + 2 * N * MV * MA + N * MV * MH +
                   N * MV * S * (MA + 1) * 2 + N * MV * S * G + N * MV * S + 64, 0);
    cx_i.assign(4 * N * MV, 0);
    cx_f.assign(4 * N * MV, 0.f);
    cx_d.assign(6 * N * MV, 0.0);
    cx.seq_cx_i = cx_i.data(); cx.seq_cx_f = cx_f.data(); cx.seq_cx_d = cx_d.data(); cx.graph_cx = cx_d.data() + 3 * N * MV;
    f64.assign(4 * N * MC + 6 * N * MH + N * MV, 0.0);
    u8.assign(N * MH * ML + N * MV * MH + N * MP, 0);
    i32.assign(2 * N * MV * MA, 0);
    uint32_t* u = u32.data();
    double* d = f64.data();
    uint8_t* b = u8.data();
    int32_t* s = i32.data();
    auto tu = [&](size_t c) { uint32_t* r = u; u += c; return r; };
    auto td = [&](size_t c) { double* r = d; d += c; return r; };
    auto tb = [&](size_t c) { uint8_t* r = b; b += c; return r; };
    auto ti = [&](size_t c) { int32_t* r = s; s += c; return r; };
    gate.max_approx = tu(N); gate.max_exact = tu(N);
    asmb.win_status = tu(N); asmb.win_k = tu(N); asmb.win_ncomp = tu(N);
    asmb.comp_anchor = tu(N * MC); asmb.comp_hap0 = tu(N * MC); asmb.comp_nhaps = tu(N * MC);
    asmb.comp_cx = tu(3 * N * MC); asmb.comp_cxf = td(4 * N * MC);
    asmb.hap_len = tu(N * MH); asmb.hap_nruns = tu(N * MH); asmb.hap_stats = td(6 * N * MH);
    asmb.hap_bases = tb(N * MH * ML); asmb.hap_runs = tu(2 * N * MH * MR);
    vars.win_nvars = tu(N); vars.var_comp = tu(N * MV); vars.var_pos = tu(N * MV); vars.var_ref_start = tu(N * MV);
    vars.var_ref_off = tu(N * MV); vars.var_ref_len = tu(N * MV); vars.var_nalts = tu(N * MV);
    vars.alt_off = tu(N * MV * MA); vars.alt_len = tu(N * MV * MA); vars.alt_type = ti(N * MV * MA);
    vars.alt_length = ti(N * MV * MA); vars.var_hap_allele = tb(N * MV * MH); vars.var_hap_start = tu(N * MV * MH);
    vars.allele_pool = tb(N * MP);
    geno.allele_counts = tu(N * MV * S * (MA + 1) * 2); geno.var_qual = td(N * MV);
    geno.var_pl = tu(N * MV * S * G); geno.var_gq = tu(N * MV * S);  // FORMAT PL / GQ (and GT = the smallest PL)
  }
};

template <class T>
class Channel {  // bounded hand-over between two stages
 public:
  explicit Channel(size_t cap) : cap_(cap) {}
  void Push(T v) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return q_.size() < cap_; });
    q_.push_back(std::move(v));
    cv_.notify_all();
  }
  void Close() {
    std::lock_guard<std::mutex> lk(mu_);
    closed_ = true;
    cv_.notify_all();
  }
  bool Pop(T* out) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    *out = std::move(q_.front());
    q_.pop_front();
    cv_.notify_all();
    return true;
  }
  bool Peek(T** out) {  // the element a later Pop will return, once there is one (or the channel is closed and empty)
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    *out = &q_.front();
    return true;
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<T> q_;
  size_t cap_;
  bool closed_ = false;
};

struct Job {
  std::unique_ptr<FlatBatch> batch;
  std::unique_ptr<Outputs> out;
  int rc = MA_OK;
  std::string err;
};

void DumpBatch(const std::string& dir, FlatBatch const& fb) {
  auto put = [&](const char* name, const void* p, size_t bytes) {
    FILE* f = std::fopen((dir + "/" + name).c_str(), "wb");
    if (!f) { std::perror(name); std::exit(4); }
    std::fwrite(p, 1, bytes, f);
    std::fclose(f);
  };
  put("ref_bases.u8", fb.ref_bases.data(), fb.ref_bases.size());
  put("ref_off.u32", fb.ref_off.data(), 4 * fb.ref_off.size());
  put("read_win_off.u32", fb.read_win_off.data(), 4 * fb.read_win_off.size());
  put("read_off.u64", fb.read_off.data(), 8 * fb.read_off.size());
  put("read_bases.u8", fb.read_bases.data(), fb.read_bases.size());
  put("read_quals.u8", fb.read_quals.data(), fb.read_quals.size());
  put("read_qname_id.u32", fb.read_qname_id.data(), 4 * fb.read_qname_id.size());
  put("read_sample.u8", fb.read_sample.data(), fb.read_sample.size());
  put("read_flags.u8", fb.read_flags.data(), fb.read_flags.size());
  put("read_hint.i32", fb.read_hint.data(), 4 * fb.read_hint.size());
  std::vector<uint64_t> wins;
  for (auto const& w : fb.windows) {
    wins.push_back(static_cast<uint64_t>(w.chrom));
    wins.push_back(w.start1);
    wins.push_back(w.end1);
    wins.push_back(w.genome_index);
  }
  put("windows.u64", wins.data(), 8 * wins.size());
}

AlignmentSource LoadAlignments(const std::string& path, Reference const& ref) {
  bool const bam = path.size() > 4 && path.substr(path.size() - 4) == ".bam";
#ifdef LANCET2_AMD_WITH_ZLIB
  if (bam) {
    // a .bai next to the file: region-indexed access (only the blocks of the regions asked for are read and inflated);
    // without one the whole file is read once
    for (std::string const& bai : {path + ".bai", path.substr(0, path.size() - 4) + ".bai"}) {
      std::ifstream probe(bai, std::ios::binary);
      if (probe && !getenv("PIPELINE_NO_INDEX")) return AlignmentSource::OpenIndexedBam(path, bai, ref);
    }
    return AlignmentSource::LoadBam(path, ref);
  }
#else
  if (bam) throw std::runtime_error("built without zlib (-DLANCET2_AMD_WITH_ZLIB -lz): BAM input is not available");
#endif
  return AlignmentSource::LoadSam(path, ref);
}

}  // namespace

int main(int argc, char** argv) {
  // The extract stage allocates a window's flat arrays (100-200 KB each) on its collector threads and frees them on the batching
  // thread.  glibc serves blocks of that size with mmap / munmap -- a system call, fresh page faults and the process-wide
  // address-space lock per array, which is what kept eight collectors from being faster than one.  From the heap arenas
  // (one per thread) the same blocks are recycled without leaving user space.  (The reference links mimalloc for the same reason.)
  // (64-bit glibc caps M_MMAP_THRESHOLD at HEAP_MAX_SIZE / 2 = 32 MiB and REJECTS anything larger, leaving the 128 KiB default
  //  in force -- ADVICE r5; the flat arrays are far below 32 MiB)
  if (mallopt(M_MMAP_THRESHOLD, 32 << 20) != 1 || mallopt(M_TRIM_THRESHOLD, 1 << 30) != 1 || mallopt(M_TOP_PAD, 64 << 20) != 1)
    std::fprintf(stderr, "pipeline_driver: mallopt rejected a setting; large arrays fall back to mmap (slower, still correct)\n");
  std::string ref_path, out_path, dump_dir, vcf_path, command_line;
  for (int i = 0; i < argc; ++i) command_line += std::string(i ? " " : "") + argv[i];
  double gc_frac = 0.41;  // --genome-gc-bias of the reference CLI: background GC of the LongdustQ null model
  std::vector<std::string> normals, tumors, regions;
  std::string bed_path;
  WindowBuilder::Params wp;
  ReadCollector::Params rp;
  ma_params_t prm;
  ma_default_params(&prm);
  bool no_active_region = false, extract_only = false, collect_reads = false;
  int batch_windows = 512;
  int extract_threads = static_cast<int>(std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
  for (int i = 1; i < argc; ++i) {
    std::string const a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--reference" || a == "-r") ref_path = next();
    else if (a == "--normal" || a == "-n") normals.emplace_back(next());
    else if (a == "--tumor" || a == "-t") tumors.emplace_back(next());
    else if (a == "--region" || a == "-R") regions.emplace_back(next());
    else if (a == "--bed-file" || a == "-b") bed_path = next();  // cli_interface.cpp: regions from a BED file (core/bed_parser.cpp)
    else if (a == "--out" || a == "-o") out_path = next();
    else if (a == "--out-vcf") vcf_path = next();  // VCF text instead of the TSV lines (adds the SEQ_CX / GRAPH_CX annotation)
    else if (a == "--genome-gc-bias") gc_frac = std::atof(next());
    else if (a == "--window-size" || a == "-w") wp.window_length = static_cast<uint32_t>(std::atoi(next()));
    else if (a == "--pct-overlap" || a == "-p") wp.percent_overlap = static_cast<uint32_t>(std::atoi(next()));
    else if (a == "--padding" || a == "-P") wp.region_padding = static_cast<uint32_t>(std::atoi(next()));
    else if (a == "--min-kmer" || a == "-k") prm.min_k = std::atoi(next());
    else if (a == "--max-kmer" || a == "-K") prm.max_k = std::atoi(next());
    else if (a == "--kmer-step") prm.k_step = std::atoi(next());
    else if (a == "--min-anchor-cov") prm.min_anchor_cov = std::atoi(next());
    else if (a == "--min-node-cov") prm.min_node_cov = std::atoi(next());
    else if (a == "--max-sample-cov") rp.max_sample_cov = std::atof(next());
    else if (a == "--extract-pairs") rp.extract_pairs = true;
    else if (a == "--no-active-region") no_active_region = true;
    else if (a == "--batch-windows") batch_windows = std::max(1, std::atoi(next()));
    else if (a == "--extract-threads") extract_threads = std::max(1, std::atoi(next()));  // (the reference: -T, one collector per worker)
    else if (a == "--dump") dump_dir = next();
    else if (a == "--collect-reads") collect_reads = true;  // the reference-shaped collector (a Read per alignment) instead of CollectFlat
    else if (a == "--extract-only") extract_only = true;  // stage 1 alone (with --dump): no device needed
    else { std::fprintf(stderr, "pipeline_driver: unknown option %s\n", a.c_str()); return 2; }
  }
  if (ref_path.empty() || (normals.empty() && tumors.empty())) {
    std::fprintf(stderr, "usage: pipeline_driver --reference ref.fa [--normal n.sam|bam]... [--tumor t.sam|bam]... [--region chr:a-b]... [--out calls.tsv]\n");
    return 2;
  }
  Reference const ref = Reference::LoadFasta(ref_path);
  std::vector<AlignmentSource> sources;
  sources.reserve(normals.size() + tumors.size());
  std::vector<SampleInfo> samples;
  auto add_sample = [&](const std::string& path, Tag tag) {
    sources.push_back(LoadAlignments(path, ref));
    std::string name = path.substr(path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1);
    name = name.substr(0, name.find_last_of('.'));
    samples.push_back({name, tag, nullptr, 0, 0, 0});
  };
  try {  // (an eager source -- SAM text, an unindexed BAM -- is decoded here: a corrupt record ends the run as it does on a collector thread)
    for (auto const& p : normals) add_sample(p, Tag::CTRL);
    for (auto const& p : tumors) add_sample(p, Tag::CASE);
  } catch (std::exception const& e) {
    std::fprintf(stderr, "pipeline_driver: %s\n", e.what());
    return 5;
  }
  for (size_t i = 0; i < samples.size(); ++i) samples[i].source = &sources[i];
  prm.num_samples = static_cast<int32_t>(samples.size());
  prm.case_ctrl_mode = (!normals.empty() && !tumors.empty()) ? 1 : 0;  // read_collector.cpp:88-94
  if (wp.window_length + 1 > static_cast<uint32_t>(prm.max_hap_len) - 16) prm.max_hap_len = 3072;  // -w 2500 (cli maximum)

  WindowBuilder wb(&ref, wp);
  std::vector<RegionSpec> bed_regions;
  if (!bed_path.empty()) {
    try {
      bed_regions = ParseBedFile(bed_path, [&](std::string const& c) { return ref.Find(c) >= 0; });
    } catch (std::exception const& e) {
      std::fprintf(stderr, "pipeline_driver: %s\n", e.what());
      return 2;
    }
  }
  if (regions.empty() && bed_regions.empty()) wb.AddAllReferenceRegions();
  for (auto const& r : regions) wb.AddRegion(r);
  for (auto const& r : bed_regions) wb.AddRegion(r);
  std::vector<Window> const windows = wb.BuildWindows();

  ma_ctx_t* ctx = nullptr;
  int const crc = extract_only ? MA_OK : ma_create(&prm, 0, MA_MEM_HOST, &ctx);
  if (crc != MA_OK) {
    std::fprintf(stderr, "pipeline_driver: ma_create failed with %d%s\n", crc,
                 crc == MA_ERR_NO_DEVICE ? " (no HIP device: the engine has no CPU fallback)" : "");
    return 3;
  }

  Channel<Job> to_engine(3), to_flush(3);
  // batches whose records are flushed go back to the extract stage with their arrays' memory (a fresh batch is 300 MB of
  // pages to fault in, on the copying threads: a quarter of the stage's time)
  std::mutex pool_mu;
  std::vector<std::unique_ptr<FlatBatch>> batch_pool;
  auto recycle = [&](std::unique_ptr<FlatBatch> b) {
    if (!b) return;
    b->Clear();
    std::lock_guard<std::mutex> g(pool_mu);
    if (batch_pool.size() < 6) batch_pool.push_back(std::move(b));
  };
  size_t n_skipped[5] = {0, 0, 0, 0, 0};
  // busy time of each stage (what it spends on its own work, not waiting for its neighbours): windows / busy second is the
  // rate the stage could sustain alone -- the slowest of the three bounds the pipeline
  using Clock = std::chrono::steady_clock;
  auto secs = [](Clock::duration d) { return std::chrono::duration<double>(d).count(); };
  double busy_extract = 0.0, busy_engine = 0.0, busy_flush = 0.0;
  size_t n_shipped = 0;
  auto const t_start = Clock::now();
  // ---- stage 1: extract ----
  // N worker threads take windows off a shared counter (gates + read collection, each with its own ReadCollector and its own
  // handle on an indexed BAM, as the reference's workers have: async_worker.h:45); one assembler thread appends their
  // results to the batches IN WINDOW ORDER, so the batches do not depend on N.
  struct Slot {
    std::atomic<int> ready{0};
    WindowStatus st = WindowStatus::RUN;
    std::unique_ptr<FlatBatch> flat;  // the window's reads as the batch holds them (flattened on the collector's thread)
  };
  std::vector<Slot> slots(windows.size());
  std::atomic<size_t> next_window{0}, consumed{0};
  constexpr size_t kRunAhead = 4096;  // windows collected but not yet batched (bounds the memory held in slots)
  std::vector<double> worker_busy(static_cast<size_t>(extract_threads), 0.0);
  std::atomic<bool> worker_failed{false};
  std::exception_ptr worker_error;
  std::mutex worker_mu;
  // Round 6: the ordered thread no longer copies the windows' arrays into the batch (43 us of ONE thread per window: the stage's
  // bound at 23-30 k windows/s whatever the number of collectors).  It appends a window's scalars (FlatBatch::PlaceHeader), and
  // when a batch is closed and sized its windows are copied into their places by whoever has a hand free -- the collectors
  // between two windows, the ordered thread while it waits.
  struct CopyTask {
    FlatBatch* dst = nullptr;
    std::unique_ptr<FlatBatch> src;
    FlatBatch::Place place{};
    std::atomic<size_t>* left = nullptr;
  };
  std::mutex copy_mu;
  std::deque<CopyTask> copy_q;
  std::atomic<size_t> copy_pending{0};
  std::atomic<bool> extract_done{false};
  auto serve_copy = [&]() -> bool {
    if (copy_pending.load(std::memory_order_acquire) == 0) return false;
    CopyTask t;
    {
      std::lock_guard<std::mutex> g(copy_mu);
      if (copy_q.empty()) return false;
      t = std::move(copy_q.front());
      copy_q.pop_front();
    }
    copy_pending.fetch_sub(1, std::memory_order_acq_rel);
    t.dst->CopyPlaced(*t.src, t.place);
    t.src.reset();
    t.left->fetch_sub(1, std::memory_order_acq_rel);
    return true;
  };
  std::vector<std::thread> workers;
  for (int t = 0; t < extract_threads; ++t)
    workers.emplace_back([&, t] {
      std::vector<SampleInfo> mine = samples;
      std::vector<AlignmentSource> clones;
#ifdef LANCET2_AMD_WITH_ZLIB
      clones.reserve(mine.size());
      for (auto& sm : mine)
        if (sm.source->indexed() && t > 0) {
          clones.push_back(sm.source->CloneIndexed());
          sm.source = &clones.back();
        }
#endif
      ReadCollector collector(rp, mine);
      while (true) {
        if (serve_copy()) continue;  // a closed batch waits for its windows' arrays: those first
        size_t const i = next_window.load() < windows.size() ? next_window.fetch_add(1) : windows.size();
        if (i >= windows.size()) {
          if (extract_done.load(std::memory_order_acquire)) break;
          std::this_thread::sleep_for(std::chrono::microseconds(50));
          continue;
        }
        if (worker_failed.load(std::memory_order_acquire)) {  // another collector hit a corrupt input: hand over empty slots, stop
          slots[i].st = WindowStatus::SKIPPED_NONLY_REF_BASES;
          slots[i].ready.store(1, std::memory_order_release);
          continue;
        }
        try {
        while (i > consumed.load(std::memory_order_acquire) + kRunAhead) std::this_thread::sleep_for(std::chrono::microseconds(200));
        auto const t0 = Clock::now();
        Window const& w = windows[i];
        std::string_view const seq(ref.chroms[static_cast<size_t>(w.chrom)].seq.data() + (w.start1 - 1), w.Length());
        Slot& sl = slots[i];
        sl.st = PreReadGate(seq, prm.max_k, no_active_region, collector.Samples(), w);
        if (sl.st == WindowStatus::RUN) {
          // the window's reads straight into its flat arrays (ReadCollector::CollectFlat); the reference-shaped path -- a Read
          // per alignment, then FlatBatch::Add -- when the options need it (--extract-pairs) or --collect-reads asks for it
          auto flat = std::make_unique<FlatBatch>();
          if (!collect_reads && collector.CollectFlat(w, seq, flat.get())) {
            if (CrossSampleMeanCoverage(collector.Samples(), w.Length()) < static_cast<double>(prm.min_anchor_cov)) sl.st = WindowStatus::SKIPPED_ANCHOR_COVERAGE;
            else sl.flat = std::move(flat);
          } else {
            ReadCollector::Result const rc = collector.CollectRegion(w);
            if (CrossSampleMeanCoverage(rc.samples, w.Length()) < static_cast<double>(prm.min_anchor_cov)) {
              sl.st = WindowStatus::SKIPPED_ANCHOR_COVERAGE;
            } else {
              sl.flat = std::move(flat);
              sl.flat->Add(w, seq, rc.reads, &rc.samples);
            }
          }
        }
        worker_busy[static_cast<size_t>(t)] += secs(Clock::now() - t0);
        sl.ready.store(1, std::memory_order_release);
        } catch (...) {  // (a BGZF / BAI / BAM decoding error surfaces lazily, on this thread: keep the first one for main())
          {
            std::lock_guard<std::mutex> g(worker_mu);
            if (!worker_error) worker_error = std::current_exception();
          }
          worker_failed.store(true, std::memory_order_release);
          slots[i].st = WindowStatus::SKIPPED_NONLY_REF_BASES;
          slots[i].flat.reset();
          slots[i].ready.store(1, std::memory_order_release);
        }
      }
    });
  std::thread extract([&] {
    size_t hint_bases = 0, hint_reads = 0, hint_ref = 0;  // the last batch's sizes: the next one's arrays are reserved whole
    auto fresh = [&] {
      Job j;
      {
        std::lock_guard<std::mutex> g(pool_mu);
        if (!batch_pool.empty()) {
          j.batch = std::move(batch_pool.back());
          batch_pool.pop_back();
        }
      }
      if (!j.batch) j.batch = std::make_unique<FlatBatch>();
      if (hint_reads) j.batch->Reserve(static_cast<size_t>(batch_windows), hint_ref, hint_reads, hint_bases);
      return j;
    };
    Job cur = fresh();
    size_t n_dumped = 0;
    std::vector<CopyTask> staged;  // the current batch's windows, in order
    // A closed batch is NOT waited for: its copy tasks are queued and the thread goes on with the next batch's headers; the
    // batches whose copies are through are sealed and handed to the engine in order whenever this thread comes by (waiting
    // at every close put most of a batch's 512 copies on this one thread: the collectors only look between two windows).
    struct Pending {
      Job job;
      std::unique_ptr<std::atomic<size_t>> left;
    };
    std::deque<Pending> pending;
    auto flush_ready = [&](bool wait_all) {
      while (!pending.empty()) {
        if (pending.front().left->load(std::memory_order_acquire) != 0) {
          if (!wait_all) return;
          if (!serve_copy()) std::this_thread::yield();
          continue;
        }
        auto const ts = Clock::now();
        Job j = std::move(pending.front().job);
        pending.pop_front();
        j.batch->Seal();
        busy_extract += secs(Clock::now() - ts);  // (the Push below may wait for the engine: not this stage's time)
        if (!dump_dir.empty()) {
          char sub[64];
          std::snprintf(sub, sizeof sub, "/batch_%04zu", n_dumped++);
          std::string const d = dump_dir + sub;
          std::string const cmd = "mkdir -p '" + d + "'";
          if (std::system(cmd.c_str()) != 0) std::exit(4);
          DumpBatch(d, *j.batch);
        }
        to_engine.Push(std::move(j));
      }
    };
    auto ship = [&] {
      if (cur.batch->windows.empty()) return;
      auto const ts = Clock::now();
      cur.batch->SizeForPlaced();
      hint_bases = cur.batch->read_bases.size() + cur.batch->read_bases.size() / 16;
      hint_reads = cur.batch->read_qname_id.size() + cur.batch->read_qname_id.size() / 16;
      hint_ref = cur.batch->ref_bases.size() + 64;
      auto left = std::make_unique<std::atomic<size_t>>(staged.size());
      {
        std::lock_guard<std::mutex> g(copy_mu);
        for (auto& t : staged) {
          t.dst = cur.batch.get();
          t.left = left.get();
          copy_q.push_back(std::move(t));
        }
      }
      copy_pending.fetch_add(staged.size(), std::memory_order_acq_rel);
      staged.clear();
      pending.push_back(Pending{std::move(cur), std::move(left)});
      busy_extract += secs(Clock::now() - ts);
      cur = fresh();
      flush_ready(false);
    };
    for (size_t i = 0; i < windows.size(); ++i) {
      Slot& sl = slots[i];
      while (!sl.ready.load(std::memory_order_acquire)) {  // (no spinning: the collectors need the cores)
        flush_ready(false);
        auto const tc = Clock::now();
        bool const copied = serve_copy();
        if (copied) busy_extract += secs(Clock::now() - tc);
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      auto const t0 = Clock::now();
      n_skipped[static_cast<int>(sl.st)]++;
      if (sl.st == WindowStatus::RUN) {
        CopyTask t;
        t.place = cur.batch->PlaceHeader(*sl.flat);
        t.src = std::move(sl.flat);
        staged.push_back(std::move(t));
        n_shipped++;
      }
      sl.flat.reset();
      consumed.store(i + 1, std::memory_order_release);
      busy_extract += secs(Clock::now() - t0);
      if (static_cast<int>(cur.batch->windows.size()) >= batch_windows) ship();
    }
    ship();
    flush_ready(true);
    extract_done.store(true, std::memory_order_release);
    to_engine.Close();
  });
  // ---- stage 2: engine ----
  std::thread engine([&] {
    Job j;
    while (to_engine.Pop(&j)) {
      if (extract_only) {
        recycle(std::move(j.batch));
        continue;
      }
      Job* nxt = nullptr;
      if (to_engine.Peek(&nxt)) ma_prefetch_batch(ctx, &nxt->batch->view);  // uploads under this batch's kernels
      auto const t0 = Clock::now();
      j.out = std::make_unique<Outputs>();
      j.out->Allocate(prm, j.batch->view.n_windows);
      j.rc = ma_process_batch(ctx, &j.batch->view, &j.out->gate, &j.out->asmb, &j.out->vars, &j.out->geno);
      if (j.rc == MA_OK && !vcf_path.empty())  // INFO SEQ_CX / GRAPH_CX (core/variant_builder.cpp:159-160)
        j.rc = ma_annotate_batch(ctx, &j.batch->view, &j.out->asmb, &j.out->vars, gc_frac, &j.out->cx);
      if (j.rc != MA_OK) j.err = ma_last_error(ctx);
      busy_engine += secs(Clock::now() - t0);
      to_flush.Push(std::move(j));
    }
    to_flush.Close();
  });
  // ---- stage 3: store + ordered flush ----
  bool const as_vcf = !vcf_path.empty();
  FILE* out = as_vcf ? std::fopen(vcf_path.c_str(), "w") : (out_path.empty() ? stdout : std::fopen(out_path.c_str(), "w"));
  if (!out) { std::perror("--out"); return 4; }
  std::vector<SampleInfo> sorted_samples = samples;
  SortSamples(sorted_samples);
  std::vector<Tag> tags;
  for (auto const& s_ : sorted_samples) tags.push_back(s_.tag);
  if (as_vcf) std::fputs(VcfHeader(ref, sorted_samples, prm.case_ctrl_mode != 0, true, command_line, ref_path).c_str(), out);
  VariantStore store;
  size_t n_records = 0, n_assembled = 0, n_flagged = 0, idx_to_flush = 0;
  int rc_all = 0;
  auto write = [&](std::vector<VariantRecord> recs) {
    for (auto const& r : recs)
      std::fprintf(out, "%s\n", as_vcf ? AsVcfRecord(r, ref, tags, prm.case_ctrl_mode != 0).c_str() : r.AsLine(ref).c_str());
    n_records += recs.size();
  };
  {
    Job j;
    while (to_flush.Pop(&j)) {
      if (j.rc != MA_OK) {
        std::fprintf(stderr, "pipeline_driver: batch failed (%d): %s\n", j.rc, j.err.c_str());
        rc_all = 5;
        continue;
      }
      // pipeline_executor.cpp:215-252: the flush lags the last window done by NUM_BUFFER_WINDOWS = 100 windows, so that a
      // call can still be replaced by a better covered duplicate from a window that overlaps its own (batches finish in
      // window order here; windows the gates skipped count as done)
      auto const tf = Clock::now();
      store.AddVariants(RecordsOfBatch(prm, *j.batch, j.out->vars, j.out->geno, as_vcf ? &j.out->cx : nullptr));
      size_t const done_upto = j.batch->windows.back().genome_index + 1;
      constexpr size_t kBufferWindows = 100;
      if (done_upto > kBufferWindows && done_upto - kBufferWindows > idx_to_flush) {
        idx_to_flush = done_upto - kBufferWindows;
        write(store.ExtractBeforeWindow(windows[idx_to_flush]));
      }
      for (int w = 0; w < j.batch->view.n_windows; ++w) {
        uint32_t const st = j.out->asmb.win_status[w];
        n_assembled += (st & MA_W_NO_HAPLOTYPE) ? 0 : 1;
        n_flagged += (st & ~static_cast<uint32_t>(MA_W_NO_HAPLOTYPE | MA_W_BFS_LIMIT)) ? 1 : 0;
      }
      busy_flush += secs(Clock::now() - tf);
      recycle(std::move(j.batch));
    }
    auto const tf = Clock::now();
    write(store.ExtractAll());
    busy_flush += secs(Clock::now() - tf);
  }
  for (auto& t : workers) t.join();
  extract.join();
  engine.join();
  if (ctx) ma_destroy(ctx);
  if (out != stdout) std::fclose(out);
  if (worker_error) {  // a collector thread's decoding error: the driver's own message and exit code, not std::terminate
    try {
      std::rethrow_exception(worker_error);
    } catch (std::exception const& e) {
      std::fprintf(stderr, "pipeline_driver: reading the alignments failed: %s\n", e.what());
    } catch (...) {
      std::fprintf(stderr, "pipeline_driver: reading the alignments failed\n");
    }
    return 5;
  }
  std::fprintf(stderr,
               "pipeline_driver: %zu windows (%zu N-only, %zu max-k repeat, %zu inactive, %zu below anchor coverage), %zu assembled, "
               "%zu with a capacity flag, %zu records\n",
               windows.size(), n_skipped[1], n_skipped[2], n_skipped[3], n_skipped[4], n_assembled, n_flagged, n_records);
  double const wall = secs(Clock::now() - t_start);
  size_t blocks = 0;
  bool any_indexed = false;
#ifdef LANCET2_AMD_WITH_ZLIB
  for (auto const& src : sources) {
    blocks += src.blocks_inflated();
    any_indexed = any_indexed || src.indexed();
  }
#endif
  auto rate = [](size_t nn, double t) { return t > 0.0 ? static_cast<double>(nn) / t : 0.0; };
  double collect_cpu = 0.0, collect_max = 0.0;
  for (double b : worker_busy) {
    collect_cpu += b;
    collect_max = std::max(collect_max, b);
  }
  double const busy_batching = busy_extract;
  busy_extract = std::max(busy_batching, collect_max);  // the stage's span: collectors and the ordered batching thread run side by side
  std::fprintf(stderr,
               "pipeline_driver: stages -- extract %.3f s busy (%.0f windows/s tiled, %.0f shipped/s) with %d collector thread(s), %.3f cpu-s of collection, %.3f s of ordered batching, engine %.3f s busy (%.0f windows/s), "
               "flush %.3f s busy (%.0f windows/s); wall %.3f s (%.0f shipped windows/s)%s\n",
               busy_extract, rate(windows.size(), busy_extract), rate(n_shipped, busy_extract), extract_threads, collect_cpu, busy_batching, busy_engine, rate(n_shipped, busy_engine),
               busy_flush, rate(n_shipped, busy_flush), wall, rate(n_shipped, wall),
               any_indexed ? (std::string("; indexed BAM: ") + std::to_string(blocks) + " BGZF blocks inflated").c_str() : "");
  return rc_all;
}
