// host_driver.cpp -- a C++ host playing Lancet2's per-window worker on top of the C-ABI (include/microasm.h).
//
// What a Lancet2 maintainer would write to put the MI355X engine behind the existing pipeline:
//   * Flatten(): a span of windows (reference region + collected reads, the arguments of
//     VariantBuilder::ProcessWindow, core/variant_builder.cpp:201-276) -> one ma_batch_t (struct of arrays);
//   * one ma_ctx_t and ONE FEEDER THREAD PER DEVICE, each with its own output buffers, replacing the N worker threads of
//     PipelineExecutor::Execute (core/pipeline_executor.cpp:174-197); batch j of the sorted window list goes to device
//     j mod G -- static sharding, no collective, windows never communicate (docs/guides/architecture.md:124);
//   * results leave in WINDOW ORDER whatever order the devices finish in (the ordered flush of
//     core/pipeline_executor.cpp:215-252), as VCF-like records: POS REF ALT QUAL and per-sample allele depths;
//   * windows that come back with an output-format capacity flag (more haplotypes / longer haplotypes / more variants
//     than the caller's fixed-stride buffers hold) are re-submitted once on a context with larger caps: the caller owns
//     those buffers, so only the caller can grow them (graph-internal capacities are retried inside the library).
// There is NO CPU fallback: without a HIP device ma_create fails and the driver exits with code 3.
//
// The windows are synthetic (seeded mt19937_64, seed 0x5EED5EED5EED5EED + window index, as the reference's tests seed
// theirs: tests/cbdg/kmer_test.cpp:81); with htslib on the box, ReadCollector fills `Window` instead
// (core/read_collector.cpp:106-309) and nothing below the Flatten() call changes.
//
//   g++ -std=c++17 -O2 examples/host_driver.cpp -Iinclude -Llancet2_amd -lmicroasm -Wl,-rpath,$PWD/lancet2_amd
//       -Wl,--allow-shlib-undefined -lpthread -o host_driver
//   ./host_driver --windows 256 --batch 64 --feeders 2 [--devices 1] [--germline] [--dump DIR] [--out FILE]
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "microasm.h"

namespace {

struct Read {  // cbdg::Read (cbdg/read.h:19-118) as far as the path needs it
  std::string seq, qual;
  uint32_t qname_id = 0;  // host-interned QNAME
  uint8_t sample = 0;
  bool is_case = false, reverse = false, pass = true;
  int32_t hint = MA_NO_HINT;  // StartPos0 - window start
};
struct Window {
  std::string ref;
  std::vector<Read> reads;  // collector order (core/read_collector.cpp:42-53)
};

// ---- synthetic windows ---------------------------------------------------------------------------------------------
Window MakeWindow(uint64_t index) {
  std::mt19937_64 rng(0x5EED5EED5EED5EEDULL + index);
  auto base = [&]() { return "ACGT"[rng() & 3]; };
  constexpr int W = 1001, RL = 150, FLANK = 300;
  std::string genome(W + 2 * FLANK, 'A');
  for (auto& c : genome) c = base();
  // one germline SNV on both samples' second haplotype, one somatic 3-base deletion in the tumour only
  std::string hap_b = genome, hap_t;
  size_t const snv = FLANK + 200 + rng() % 200;
  hap_b[snv] = hap_b[snv] == 'A' ? 'C' : 'A';
  size_t const del = FLANK + 550 + rng() % 200;
  hap_t = genome.substr(0, del) + genome.substr(del + 3);
  Window w;
  w.ref = genome.substr(FLANK, W);
  uint32_t qn = 0;
  auto add_reads = [&](const std::string& hap, uint8_t sample, bool is_case, int pairs) {
    for (int p = 0; p < pairs; ++p) {
      size_t const ins = 300 + rng() % 200;
      size_t const fs = rng() % (hap.size() - ins);
      for (int mate = 0; mate < 2; ++mate) {
        size_t const st = mate == 0 ? fs : fs + ins - RL;
        if (st + RL <= FLANK || st >= FLANK + W) continue;
        Read r;
        r.seq = hap.substr(st, RL);
        r.qual.assign(RL, static_cast<char>(35));
        r.qname_id = qn;
        r.sample = sample;
        r.is_case = is_case;
        r.reverse = mate == 1;
        r.hint = static_cast<int32_t>(st) - FLANK;
        w.reads.push_back(std::move(r));
      }
      ++qn;
    }
  };
  int const pairs = 30 * (W + 2 * RL) / (2 * RL) / 2;  // 30x per sample over two haplotypes
  add_reads(genome, 0, false, pairs);
  add_reads(hap_b, 0, false, pairs);
  add_reads(hap_b, 1, true, pairs / 2);
  add_reads(genome, 1, true, pairs / 2);
  add_reads(hap_t, 1, true, pairs);
  // collector comparator: pass-filter first, then tag (CTRL < CASE), sample, qname (read_collector.cpp:42-53)
  std::stable_sort(w.reads.begin(), w.reads.end(), [](const Read& a, const Read& b) {
    if (a.pass != b.pass) return a.pass;
    if (a.is_case != b.is_case) return !a.is_case;
    if (a.sample != b.sample) return a.sample < b.sample;
    return a.qname_id < b.qname_id;
  });
  return w;
}

// ---- Flatten: windows -> ma_batch_t --------------------------------------------------------------------------------
struct FlatBatch {
  std::vector<uint8_t> ref_bases, read_bases, read_quals, read_sample, read_flags;
  std::vector<uint32_t> ref_off, read_win_off, read_qname_id;
  std::vector<uint64_t> read_off;
  std::vector<int32_t> read_hint;
  ma_batch_t view{};
};
void Flatten(const Window* wins, int n, FlatBatch* fb) {
  *fb = FlatBatch{};
  fb->ref_off.push_back(0);
  fb->read_win_off.push_back(0);
  fb->read_off.push_back(0);
  for (int i = 0; i < n; ++i) {
    const Window& w = wins[i];
    fb->ref_bases.insert(fb->ref_bases.end(), w.ref.begin(), w.ref.end());
    fb->ref_off.push_back(static_cast<uint32_t>(fb->ref_bases.size()));
    for (const Read& r : w.reads) {
      fb->read_bases.insert(fb->read_bases.end(), r.seq.begin(), r.seq.end());
      fb->read_quals.insert(fb->read_quals.end(), r.qual.begin(), r.qual.end());
      fb->read_off.push_back(fb->read_bases.size());
      fb->read_qname_id.push_back(r.qname_id);
      fb->read_sample.push_back(r.sample);
      fb->read_flags.push_back(static_cast<uint8_t>((r.pass ? MA_RF_PASS : 0) | (r.is_case ? MA_RF_CASE : 0) |
                                                    (r.reverse ? MA_RF_REV : 0)));
      fb->read_hint.push_back(r.hint);
    }
    fb->read_win_off.push_back(static_cast<uint32_t>(fb->read_qname_id.size()));
  }
  // vector loads past the end stay inside the allocation
  fb->ref_bases.resize(fb->ref_bases.size() + 64, 0);
  fb->read_bases.resize(fb->read_bases.size() + 64, 0);
  fb->read_quals.resize(fb->read_quals.size() + 64, 0);
  ma_batch_t& b = fb->view;
  b.n_windows = n;
  b.n_reads = static_cast<int64_t>(fb->read_qname_id.size());
  b.ref_bases = fb->ref_bases.data();
  b.ref_off = fb->ref_off.data();
  b.read_win_off = fb->read_win_off.data();
  b.read_off = fb->read_off.data();
  b.read_bases = fb->read_bases.data();
  b.read_quals = fb->read_quals.data();
  b.read_qname_id = fb->read_qname_id.data();
  b.read_sample = fb->read_sample.data();
  b.read_flags = fb->read_flags.data();
  b.read_hint = fb->read_hint.data();
}

// ---- caller-owned output buffers of one feeder ---------------------------------------------------------------------
struct Outputs {
  std::vector<uint32_t> u32;
  std::vector<double> f64;
  std::vector<uint8_t> u8;
  std::vector<int32_t> i32;
  ma_gate_out_t gate{};
  ma_asm_out_t asmb{};
  ma_var_out_t vars{};
  ma_geno_out_t geno{};
  void Allocate(const ma_params_t& p, int n) {
    size_t const N = n, MC = p.max_comps, MH = p.max_haps, ML = p.max_hap_len, MR = p.max_runs, MV = p.max_vars,
                 MA = p.max_alts, MP = p.max_allele_bytes, S = p.num_samples, G = (MA + 1) * (MA + 2) / 2;
    size_t const n_u32 = 2 * N + 3 * N + 3 * N * MC + 3 * N * MC + 2 * N * MH + 2 * N * MH * MR + N + 5 * N * MV + 2 * N * MV +
                         2 * N * MV * MA + N * MV * MH + N * MV * S * (MA + 1) * 2 + N * MV * S * G + N * MV * S;
    u32.assign(n_u32 + n_u32 / 4 + 64, 0);  // (the formula is exact up to one spare N * max_vars term; keep slack)
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
    geno.var_pl = tu(N * MV * S * G); geno.var_gq = tu(N * MV * S);
  }
};

// one record per variant, "window_index \t pos0 \t REF \t ALT[,ALT] \t QUAL \t AD sample0 \t AD sample1 ..."
void EmitRecords(const ma_params_t& p, const Outputs& o, int first_window, int n, std::vector<std::string>* lines,
                 std::vector<int>* flagged) {
  int const MV = p.max_vars, MA = p.max_alts, S = p.num_samples, NA = MA + 1;
  for (int w = 0; w < n; ++w) {
    uint32_t const st = o.asmb.win_status[w];
    if (st & (MA_W_HAP_OVERFLOW | MA_W_LEN_OVERFLOW | MA_W_VAR_OVERFLOW | MA_W_TABLE_OVERFLOW)) flagged->push_back(first_window + w);
    const uint8_t* pool = o.vars.allele_pool + static_cast<size_t>(w) * p.max_allele_bytes;
    for (uint32_t v = 0; v < o.vars.win_nvars[w]; ++v) {
      size_t const vi = static_cast<size_t>(w) * MV + v;
      std::string line = std::to_string(first_window + w) + "\t" + std::to_string(o.vars.var_pos[vi]) + "\t";
      line.append(reinterpret_cast<const char*>(pool) + o.vars.var_ref_off[vi], o.vars.var_ref_len[vi]);
      line += "\t";
      for (uint32_t a = 0; a < o.vars.var_nalts[vi]; ++a) {
        if (a) line += ",";
        line.append(reinterpret_cast<const char*>(pool) + o.vars.alt_off[vi * MA + a], o.vars.alt_len[vi * MA + a]);
      }
      char buf[64];
      std::snprintf(buf, sizeof buf, "\t%.6f", o.geno.var_qual[vi]);
      line += buf;
      for (int s = 0; s < S; ++s) {
        line += "\t";
        for (uint32_t al = 0; al <= o.vars.var_nalts[vi]; ++al) {
          const uint32_t* c = o.geno.allele_counts + ((vi * S + s) * NA + al) * 2;
          if (al) line += ",";
          line += std::to_string(c[0] + c[1]);
        }
      }
      lines->push_back(std::move(line));
    }
  }
}

void DumpBatch(const std::string& dir, const FlatBatch& fb) {
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
}

struct BatchResult {
  std::vector<std::string> lines;
  std::vector<int> flagged;
  bool done = false;
  int rc = 0;
  std::string err;
};

}  // namespace

int main(int argc, char** argv) {
  int n_windows = 128, batch = 64, feeders = 1, devices = 1, germline = 0;
  std::string dump, out_path;
  for (int i = 1; i < argc; ++i) {
    std::string const a = argv[i];
    auto next = [&]() { return i + 1 < argc ? argv[++i] : "0"; };
    if (a == "--windows") n_windows = std::atoi(next());
    else if (a == "--batch") batch = std::atoi(next());
    else if (a == "--feeders") feeders = std::atoi(next());
    else if (a == "--devices") devices = std::atoi(next());
    else if (a == "--germline") germline = 1;
    else if (a == "--dump") dump = next();
    else if (a == "--out") out_path = next();
  }
  ma_params_t prm;
  ma_default_params(&prm);
  prm.min_k = prm.max_k = 25;  // BASELINE config; drop this line for the reference's default cascade
  prm.case_ctrl_mode = germline ? 0 : 1;

  // windows in pipeline order (WindowBuilder's sorted list, core/window_builder.cpp:113-135)
  std::vector<Window> windows(static_cast<size_t>(n_windows));
  for (int i = 0; i < n_windows; ++i) windows[static_cast<size_t>(i)] = MakeWindow(static_cast<uint64_t>(i));
  int const n_batches = (n_windows + batch - 1) / batch;
  if (!dump.empty()) {
    FlatBatch all;
    Flatten(windows.data(), n_windows, &all);
    DumpBatch(dump, all);
  }

  // one context + feeder thread per (device, feeder slot); batch j -> worker j mod G
  int const G = std::max(1, devices * feeders);
  std::vector<ma_ctx_t*> ctx(static_cast<size_t>(G), nullptr);
  for (int g = 0; g < G; ++g) {
    int const rc = ma_create(&prm, g % devices, MA_MEM_HOST, &ctx[static_cast<size_t>(g)]);
    if (rc != MA_OK) {
      std::fprintf(stderr, "host_driver: ma_create(device %d) failed with %d%s\n", g % devices, rc,
                   rc == MA_ERR_NO_DEVICE ? " (no HIP device: the engine has no CPU fallback)" : "");
      return 3;
    }
    ma_set_streams(ctx[static_cast<size_t>(g)], 1);
  }
  std::vector<BatchResult> results(static_cast<size_t>(n_batches));
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::thread> threads;
  for (int g = 0; g < G; ++g) {
    threads.emplace_back([&, g]() {
      FlatBatch fb;
      Outputs out;
      for (int j = g; j < n_batches; j += G) {
        int const w0 = j * batch, n = std::min(batch, n_windows - w0);
        Flatten(windows.data() + w0, n, &fb);
        out.Allocate(prm, n);
        BatchResult r;
        r.rc = ma_process_batch(ctx[static_cast<size_t>(g)], &fb.view, &out.gate, &out.asmb, &out.vars, &out.geno);
        if (r.rc != MA_OK) r.err = ma_last_error(ctx[static_cast<size_t>(g)]);
        else EmitRecords(prm, out, w0, n, &r.lines, &r.flagged);
        r.done = true;
        {
          std::lock_guard<std::mutex> lk(mu);
          results[static_cast<size_t>(j)] = std::move(r);
        }
        cv.notify_all();
      }
    });
  }
  // ordered flush (pipeline_executor.cpp:215-252): batch j is written once every batch before it has been
  FILE* out = out_path.empty() ? stdout : std::fopen(out_path.c_str(), "w");
  if (!out) { std::perror("--out"); return 4; }
  int rc_all = 0;
  size_t n_records = 0;
  std::vector<int> flagged;
  std::vector<std::string> pending_lines;  // records held back behind a flagged window (each starts with its window index)
  for (int j = 0; j < n_batches; ++j) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&]() { return results[static_cast<size_t>(j)].done; });
    BatchResult r = std::move(results[static_cast<size_t>(j)]);
    lk.unlock();
    if (r.rc != MA_OK) {
      std::fprintf(stderr, "host_driver: batch %d failed (%d): %s\n", j, r.rc, r.err.c_str());
      rc_all = 5;
      continue;
    }
    // a window whose fixed-stride outputs overflowed is re-submitted below with larger buffers: its truncated first-pass
    // records are held back, and everything from the first such window on waits, so that the output stays in window order
    // (a real host would re-submit at once on a spare context; this example does it after the batches)
    for (auto& l : r.lines) pending_lines.push_back(std::move(l));
    flagged.insert(flagged.end(), r.flagged.begin(), r.flagged.end());
    if (flagged.empty()) {
      for (auto const& l : pending_lines) std::fprintf(out, "%s\n", l.c_str());
      n_records += pending_lines.size();
      pending_lines.clear();
    }
  }
  for (auto& t : threads) t.join();
  // output-format capacity flags: the caller owns the fixed-stride buffers, so the caller re-submits with larger ones
  if (!flagged.empty() && rc_all == 0) {
    ma_params_t big = prm;
    big.max_haps = 32; big.max_hap_len = 4096; big.max_vars = 256; big.max_allele_bytes = 16384; big.max_runs = 512;
    ma_ctx_t* c2 = nullptr;
    for (auto& c : ctx) { ma_destroy(c); c = nullptr; }
    std::map<int, std::vector<std::string>> redo;  // window -> its records from the larger buffers
    std::set<int> redone;                            // windows whose re-submission succeeded (possibly with no record)
    if (ma_create(&big, 0, MA_MEM_HOST, &c2) == MA_OK) {
      for (int w : flagged) {
        FlatBatch fb;
        Outputs o2;
        Flatten(&windows[static_cast<size_t>(w)], 1, &fb);
        o2.Allocate(big, 1);
        if (ma_process_batch(c2, &fb.view, &o2.gate, &o2.asmb, &o2.vars, &o2.geno) == MA_OK) {
          std::vector<int> still;
          EmitRecords(big, o2, w, 1, &redo[w], &still);
          redone.insert(w);
        } else {
          std::fprintf(stderr, "host_driver: re-submission of window %d failed: %s\n", w, ma_last_error(c2));
        }
      }
      ma_destroy(c2);
    } else {
      std::fprintf(stderr, "host_driver: could not create a context with larger output buffers\n");
    }
    // a flagged window that could not be re-submitted keeps its (truncated) first-pass records, and the run says so
    for (int w : flagged)
      if (!redone.count(w)) {
        std::fprintf(stderr, "host_driver: window %d: output capacity flag, first-pass records written as they are (TRUNCATED)\n", w);
        rc_all = 6;
      }
    // splice: the held-back records in window order, a re-submitted window's records in place of its first-pass ones
    int last_w = -1;
    auto flush_redo_upto = [&](int w_excl) {
      for (auto it = redo.begin(); it != redo.end() && it->first < w_excl;) {
        for (auto const& l : it->second) std::fprintf(out, "%s\n", l.c_str());
        n_records += it->second.size();
        it = redo.erase(it);
      }
    };
    for (auto const& l : pending_lines) {
      int const w = std::atoi(l.c_str());
      if (w != last_w) flush_redo_upto(w);
      last_w = w;
      if (redone.count(w)) continue;  // replaced by the re-submitted window's records
      std::fprintf(out, "%s\n", l.c_str());
      n_records++;
    }
    flush_redo_upto(INT32_MAX);
  }
  else if (!pending_lines.empty()) {  // a failed batch: what was held back goes out as it is
    for (auto const& l : pending_lines) std::fprintf(out, "%s\n", l.c_str());
    n_records += pending_lines.size();
  }
  for (auto& c : ctx)
    if (c) ma_destroy(c);
  if (out != stdout) std::fclose(out);
  std::fprintf(stderr, "host_driver: %d windows in %d batches on %d feeder(s): %zu records, %zu windows re-submitted\n", n_windows,
               n_batches, G, n_records, flagged.size());
  return rc_all;
}
