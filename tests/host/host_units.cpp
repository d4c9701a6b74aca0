// Unit checks of the host shell (lancet2_amd/host/pipeline_host.hpp) against facts the reference states about itself.
// Test infrastructure: built and run by tests/test_pipeline_host.py with plain g++ (no GPU, no engine call).
#include <cassert>
#include <cstdio>

#include "../../lancet2_amd/host/pipeline_host.hpp"

using namespace lancet2_amd::host;

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);         \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static SamRecord Rec(const char* q, int chrom, int64_t pos0, const char* cigar, uint8_t mapq = 60, uint16_t flag = 0x3,
                     const char* md = nullptr, size_t len = 0) {
  SamRecord r;
  r.qname = q;
  r.chrom = chrom;
  r.pos0 = pos0;
  r.cigar = ParseCigar(cigar);
  r.mapq = mapq;
  r.flag = flag;
  size_t n = len;
  if (!n)
    for (auto const& c : r.cigar)
      if (c.op == 'M' || c.op == 'I' || c.op == 'S' || c.op == '=' || c.op == 'X') n += c.len;
  r.seq.assign(n, 'A');
  r.qual.assign(n, 30);
  if (md) {
    r.md = md;
    r.has_md = true;
  }
  return r;
}

int main() {
  // docs/guides/architecture.md:155-158: 1000-base windows at 20 % overlap step by 800; steps are multiples of 100
  CHECK(WindowBuilder::StepSize({1000, 500, 20}) == 800);
  CHECK(WindowBuilder::StepSize({1000, 500, 50}) == 500);
  CHECK(WindowBuilder::StepSize({2500, 500, 90}) == 300);  // ceil(250 / 100) * 100
  Reference ref;
  ref.chroms.push_back({"chr1", std::string(10000, 'A')});
  ref.chroms.push_back({"chrUn_x", std::string(500, 'C')});
  ref.chroms.push_back({"chr2", std::string(3000, 'G')});
  {  // window_builder.cpp:287-323
    WindowBuilder wb(&ref, {1000, 500, 20});
    RegionSpec r = RegionSpec::Parse("chr1:3000-5000");
    wb.PadInputRegion(r);
    CHECK(*r.start == 2500 && *r.end == 5500);
    RegionSpec lo = RegionSpec::Parse("chr1:100-9800");
    wb.PadInputRegion(lo);
    CHECK(*lo.start == 1 && *lo.end == 10000);  // start underflows, end within the padding of the contig end
    RegionSpec tiny = RegionSpec::Parse("chr1:5000-5010");  // shorter than a window even when padded: grown around itself
    WindowBuilder wb0(&ref, {1000, 0, 20});
    wb0.PadInputRegion(tiny);
    CHECK(tiny.Length() >= 1000 && *tiny.start < 5000 && *tiny.end > 5010);
  }
  {  // tiling (window_builder.cpp:140-205): closed 1001-base windows every 800 bases while start + 1000 <= region end
    WindowBuilder wb(&ref, {1000, 500, 20});
    wb.AddRegion("chr1:1-6000");
    wb.AddRegion("chr1:1-6000");  // duplicate regions yield duplicate windows: removed
    auto const w = wb.BuildWindows();
    CHECK(w.size() == 7);
    for (size_t i = 0; i < w.size(); ++i) {
      CHECK(w[i].start1 == 1 + 800 * i && w[i].end1 == w[i].start1 + 1000 && w[i].Length() == 1001);
      CHECK(w[i].genome_index == i);
    }
    WindowBuilder all(&ref, {1000, 500, 20});
    all.AddAllReferenceRegions();  // chrUn_* is excluded (window_builder.cpp:41-53)
    auto const wa = all.BuildWindows();
    for (auto const& x : wa) CHECK(x.chrom != 1);
    CHECK(wa.front().chrom == 0 && wa.back().chrom == 2);
    CHECK(WindowBuilder::ShouldExcludeChrom("chrM") && WindowBuilder::ShouldExcludeChrom("chr1_KI270706v1_random") &&
          !WindowBuilder::ShouldExcludeChrom("chr10"));
  }
  {  // comparator (read_collector.cpp:42-53)
    Read a, b;
    a.passes = false; b.passes = true;
    CHECK(CompareReadsByPriority(b, a) && !CompareReadsByPriority(a, b));  // pass first
    a.passes = true; a.tag = Tag::CASE; b.tag = Tag::CTRL;
    CHECK(CompareReadsByPriority(b, a));                                   // CTRL (2) before CASE (4)
    a.tag = Tag::CTRL; a.sample_name = "n2"; b.sample_name = "n1";
    CHECK(CompareReadsByPriority(b, a));
    a.sample_name = "n1"; a.qname = "q2"; b.qname = "q10";
    CHECK(CompareReadsByPriority(b, a));                                   // "q10" < "q2" as strings
    a.qname = "q10"; a.start0 = 5; b.start0 = 4;
    CHECK(CompareReadsByPriority(b, a));
  }
  {  // active region: two reads with a quality >= 20 mismatch at one genome position (active_region_detector.cpp:85-126)
    CountMap m;
    std::vector<uint8_t> q(20, 30);
    CHECK(!ParseMd("10A9", q, 100, &m));
    CHECK(ParseMd("10C9", q, 100, &m));   // second hit at genome position 110
    CountMap m2;
    std::vector<uint8_t> lowq(20, 10);
    CHECK(!ParseMd("10A9", lowq, 100, &m2) && !ParseMd("10A9", lowq, 100, &m2));  // low base quality: not counted
    CountMap m3;
    CHECK(!ParseMd("5^A15", q, 100, &m3) && m3.size() == 1);  // deletion bases are letters too: the reference counts them,
    CountMap m4;                                               // without advancing -- so one read with a two-base deletion
    CHECK(ParseMd("5^AC15", q, 100, &m4));                     // reaches the threshold by itself (restated as is)
    AlignmentSource src;
    src.recs.push_back(Rec("a", 0, 1000, "50M2I48M"));
    src.recs.push_back(Rec("b", 0, 1010, "40M2I58M"));  // both insertions sit at genome position 1050
    src.recs.push_back(Rec("c", 0, 3000, "100M", 0));   // mapq 0: ignored
    src.Finish(ref.chroms.size());
    std::vector<SampleInfo> ss{{"s", Tag::CTRL, &src, 0, 0, 0}};
    CHECK(IsActiveRegion(ss, Window{0, 801, 1801, 0}));
    CHECK(!IsActiveRegion(ss, Window{0, 2401, 3401, 0}));
    AlignmentSource clip;
    clip.recs.push_back(Rec("a", 0, 500, "20S80M"));
    clip.recs.push_back(Rec("b", 0, 500, "10S90M"));  // both clips at genome position 500
    clip.Finish(ref.chroms.size());
    std::vector<SampleInfo> cs{{"s", Tag::CTRL, &clip, 0, 0, 0}};
    CHECK(IsActiveRegion(cs, Window{0, 1, 1001, 0}));
  }
  {  // collector: filters, both mates of a pair kept or dropped together, coverage cap, deterministic
    AlignmentSource src;
    for (int i = 0; i < 400; ++i) {
      std::string const q = "p" + std::to_string(i);
      src.recs.push_back(Rec(q.c_str(), 0, 1000 + i, "100M", 60, 0x63));
      src.recs.push_back(Rec(q.c_str(), 0, 1300 + i, "100M", 60, 0x93));
    }
    src.recs.push_back(Rec("dup", 0, 1200, "100M", 60, 0x400 | 0x3));
    src.recs.push_back(Rec("lowq", 0, 1200, "100M", 5));
    src.recs.push_back(Rec("qcfail", 0, 1200, "100M", 60, 0x200 | 0x3));
    src.Finish(ref.chroms.size());
    ReadCollector::Params p;
    p.max_sample_cov = 20.0;  // 20x over 1001 bases = 20 020 bases = 201 reads of 100
    ReadCollector rc(p, {{"s", Tag::CASE, &src, 0, 0, 0}});
    Window const w{0, 801, 1801, 0};
    auto const r1 = rc.CollectRegion(w);
    auto const r2 = rc.CollectRegion(w);
    CHECK(r1.reads.size() == r2.reads.size() && !r1.reads.empty());
    std::unordered_map<std::string, int> per;
    for (size_t i = 0; i < r1.reads.size(); ++i) {
      CHECK(r1.reads[i].qname == r2.reads[i].qname && r1.reads[i].start0 == r2.reads[i].start0);
      CHECK(r1.reads[i].qname[0] == 'p');  // duplicate / low mapq / QC-fail records never get in
      per[r1.reads[i].qname]++;
      if (i) CHECK(!CompareReadsByPriority(r1.reads[i], r1.reads[i - 1]));
    }
    for (auto const& kv : per) CHECK(kv.second == 2);  // pairs travel together
    CHECK(r1.samples[0].sampled_reads == 201);          // ceil(20 020 / 100) of the 800 passing reads
    CHECK(r1.reads.size() >= 201 && r1.reads.size() <= 402);
    CHECK(CrossSampleMeanCoverage(r1.samples, w.Length()) > 19.0);
    // CollectFlat: the same window straight into flat arrays -- byte for byte what CollectRegion + FlatBatch::Add append,
    // in the coverage-capped case (above) and in the uncapped one, with a second sample of the other tag beside the first
    for (double cap : {20.0, 1000.0}) {
      ReadCollector::Params pc;
      pc.max_sample_cov = cap;
      ReadCollector ca(pc, {{"s", Tag::CASE, &src, 0, 0, 0}, {"n", Tag::CTRL, &src, 0, 0, 0}});
      ReadCollector cb(pc, {{"s", Tag::CASE, &src, 0, 0, 0}, {"n", Tag::CTRL, &src, 0, 0, 0}});
      std::string const seq(w.Length(), 'A');
      auto const rr = ca.CollectRegion(w);
      FlatBatch fa, fbb;
      fa.Add(w, seq, rr.reads, &rr.samples);
      CHECK(cb.CollectFlat(w, seq, &fbb));
      CHECK(fa.read_bases == fbb.read_bases && fa.read_quals == fbb.read_quals && fa.read_off == fbb.read_off);
      CHECK(fa.read_qname_id == fbb.read_qname_id && fa.read_sample == fbb.read_sample && fa.read_flags == fbb.read_flags);
      CHECK(fa.read_hint == fbb.read_hint && fa.read_win_off == fbb.read_win_off && fa.ref_bases == fbb.ref_bases && fa.ref_off == fbb.ref_off);
      CHECK(fa.sample_cov == fbb.sample_cov && !fa.read_qname_id.empty());
      CHECK(ca.Samples()[0].sampled_bases == cb.Samples()[0].sampled_bases && ca.Samples()[1].sampled_reads == cb.Samples()[1].sampled_reads);
    }
  }
  {  // gates (variant_builder.cpp:107-132)
    std::vector<SampleInfo> none;
    CHECK(PreReadGate(std::string(1001, 'N'), 127, true, none, Window{0, 1, 1001, 0}) == WindowStatus::SKIPPED_NONLY_REF_BASES);
    std::string rep(1001, 'A');
    CHECK(PreReadGate(rep, 127, true, none, Window{0, 1, 1001, 0}) == WindowStatus::SKIPPED_REF_REPEAT_SEEN);
    CHECK(!HasExactRepeat("ACGTACGA", 5) && HasExactRepeat("ACGTAACGTA", 5));
  }
  {  // store (variant_store.cpp:20-79): same CHROM + POS + REF keeps the better covered call; ordered extraction
    VariantStore st;
    VariantRecord a{0, 1500, "A", {"T"}, 10.0, {{10, 0}, {8, 5}}, 1};
    VariantRecord b{0, 1500, "A", {"T", "G"}, 12.0, {{12, 0, 0}, {9, 6, 1}}, 2};  // the overlapping window saw more
    VariantRecord c{0, 900, "AC", {"A"}, 5.0, {{5, 0}, {5, 0}}, 0};               // no ALT support: never written
    VariantRecord d{2, 10, "G", {"C"}, 7.0, {{3, 0}, {2, 2}}, 9};
    VariantRecord e{0, 1500, "A", {"T"}, 9.0, {{1, 0}, {1, 1}}, 3};               // worse covered duplicate: ignored
    st.AddVariants({a, c, d});
    st.AddVariants({b, e});
    CHECK(st.Size() == 3);
    auto const first = st.ExtractBeforeWindow(Window{0, 1601, 2601, 2});  // POS < the window's END
    CHECK(first.size() == 1 && first[0].alts.size() == 2 && first[0].window_index == 2);
    auto const rest = st.ExtractAll();
    CHECK(rest.size() == 1 && rest[0].chrom == 2 && st.Size() == 0);
  }
  {  // FlatBatch::Append (the extract stage's workers flatten a window each, the ordered assembler appends): the batch is the
    // one that Add() on the same windows in the same order builds -- arrays, offsets, per-window name ids and coverage.
    auto mk = [](const char* qn, const char* seq, uint8_t q0, int64_t start0, uint16_t flag, Tag tag, size_t sidx, uint32_t clip) {
      Read r;
      r.qname = qn; r.seq = seq; r.qual.assign(r.seq.size(), q0); r.start0 = start0; r.chrom = 0; r.flag = flag; r.tag = tag;
      r.sample_index = sidx; r.leading_clip = clip; r.passes = q0 != 7;
      return r;
    };
    std::vector<Read> r0 = {mk("a", "ACGTAC", 30, 104, 0x63, Tag::CTRL, 0, 0), mk("a", "TTGACA", 0xFF, 180, 0x93, Tag::CTRL, 0, 2),
                            mk("b", "GGGTTT", 7, 150, 0x10, Tag::CASE, 1, 0)};
    std::vector<Read> r1 = {};
    std::vector<Read> r2 = {mk("b", "CATCAT", 20, 905, 0, Tag::CASE, 1, 1), mk("c", "AAAAAC", 25, 950, 0x10, Tag::CTRL, 0, 0)};
    std::vector<SampleInfo> si(2);
    si[0].sampled_bases = 1200; si[1].sampled_bases = 600;
    Window const w0{0, 101, 1101, 0}, w1{0, 501, 1501, 1}, w2{0, 901, 1901, 2};
    std::string const ref(1001, 'A');
    FlatBatch serial;
    serial.Add(w0, ref, r0, &si); serial.Add(w1, ref, r1, &si); serial.Add(w2, ref, r2, &si);
    FlatBatch pieces;
    for (auto const& pr : {std::make_pair(&w0, &r0), std::make_pair(&w1, &r1), std::make_pair(&w2, &r2)}) {
      FlatBatch one;
      one.Add(*pr.first, ref, *pr.second, &si);
      pieces.Append(one);
    }
    {  // round 6: the same batch placed in two steps (headers in order, arrays copied afterwards in ANY order)
      FlatBatch placed;
      std::vector<std::pair<std::unique_ptr<FlatBatch>, FlatBatch::Place>> todo;
      for (auto const& pr : {std::make_pair(&w0, &r0), std::make_pair(&w1, &r1), std::make_pair(&w2, &r2)}) {
        auto one = std::make_unique<FlatBatch>();
        one->Add(*pr.first, ref, *pr.second, &si);
        FlatBatch::Place const pl = placed.PlaceHeader(*one);
        todo.emplace_back(std::move(one), pl);
      }
      placed.SizeForPlaced();
      for (size_t i = todo.size(); i-- > 0;) placed.CopyPlaced(*todo[i].first, todo[i].second);
      CHECK(serial.ref_bases == placed.ref_bases && serial.read_bases == placed.read_bases && serial.read_quals == placed.read_quals);
      CHECK(serial.read_sample == placed.read_sample && serial.read_flags == placed.read_flags && serial.read_hint == placed.read_hint);
      CHECK(serial.ref_off == placed.ref_off && serial.read_off == placed.read_off && serial.read_win_off == placed.read_win_off);
      CHECK(serial.read_qname_id == placed.read_qname_id && serial.sample_cov == placed.sample_cov && placed.windows.size() == 3);
      // a recycled batch (Clear() keeps the arrays' memory): built again from two of the windows it equals a fresh one
      placed.Seal();
      placed.Clear();
      CHECK(placed.windows.empty() && placed.read_bases.empty() && placed.ref_off == (std::vector<uint32_t>{0}) && placed.view.n_windows == 0);
      FlatBatch fresh2;
      fresh2.Add(w2, ref, r2, &si); fresh2.Add(w0, ref, r0, &si);
      std::vector<std::pair<std::unique_ptr<FlatBatch>, FlatBatch::Place>> again;
      for (auto const& pr : {std::make_pair(&w2, &r2), std::make_pair(&w0, &r0)}) {
        auto one = std::make_unique<FlatBatch>();
        one->Add(*pr.first, ref, *pr.second, &si);
        FlatBatch::Place const pl = placed.PlaceHeader(*one);
        again.emplace_back(std::move(one), pl);
      }
      placed.SizeForPlaced();
      for (auto const& t : again) placed.CopyPlaced(*t.first, t.second);
      CHECK(fresh2.read_bases == placed.read_bases && fresh2.read_quals == placed.read_quals && fresh2.read_off == placed.read_off);
      CHECK(fresh2.read_win_off == placed.read_win_off && fresh2.ref_off == placed.ref_off && fresh2.read_hint == placed.read_hint);
      CHECK(fresh2.read_qname_id == placed.read_qname_id && fresh2.read_flags == placed.read_flags && fresh2.ref_bases == placed.ref_bases);
    }
    CHECK(serial.ref_bases == pieces.ref_bases && serial.read_bases == pieces.read_bases && serial.read_quals == pieces.read_quals);
    CHECK(serial.read_sample == pieces.read_sample && serial.read_flags == pieces.read_flags && serial.read_hint == pieces.read_hint);
    CHECK(serial.ref_off == pieces.ref_off && serial.read_off == pieces.read_off && serial.read_win_off == pieces.read_win_off);
    CHECK(serial.read_qname_id == pieces.read_qname_id && serial.sample_cov == pieces.sample_cov && serial.windows.size() == 3);
    CHECK(serial.read_win_off == (std::vector<uint32_t>{0, 3, 3, 5}) && serial.read_qname_id == (std::vector<uint32_t>{0, 0, 1, 0, 1}));
    CHECK(serial.read_quals[6] == 0 && serial.read_hint[0] == 4 && serial.read_hint[1] == 78);  // 0xFF qualities -> 0; hint = start - window start - clip
  }
  {  // core/variant_builder.cpp:184-199 BEFORE core/variant_store.cpp:20-42: a call without ALT support never reaches the store.
    // Two overlapping windows report the same CHROM + POS + REF; window 0's call has MORE total coverage but no ALT read,
    // window 1's has less coverage and ALT support.  Filtering at the flush only, the unsupported call would win the
    // keep-better-covered rule and the site would vanish from the output.
    ma_params_t p{};
    p.max_vars = 2; p.max_alts = 2; p.num_samples = 2; p.max_allele_bytes = 8;
    FlatBatch fb;
    fb.windows = {Window{0, 1, 1001, 0}, Window{0, 801, 1801, 1}};
    int const MV = 2, MA = 2, S = 2, NA = 3;
    std::vector<uint8_t> pool = {'A', 'T', 0, 0, 0, 0, 0, 0, 'A', 'T', 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> win_nvars = {1, 1}, var_pos(2 * MV, 0), var_ref_off(2 * MV, 0), var_ref_len(2 * MV, 1), var_nalts(2 * MV, 1);
    std::vector<uint32_t> alt_off(2 * MV * MA, 1), alt_len(2 * MV * MA, 1);
    std::vector<int32_t> alt_type(2 * MV * MA, 0), alt_length(2 * MV * MA, 1);
    var_pos[0] = 899;          // window 0 starts at 1   -> POS 900
    var_pos[1 * MV] = 99;      // window 1 starts at 801 -> POS 900
    std::vector<double> var_qual(2 * MV, 0.0);
    var_qual[1 * MV] = 33.0;
    std::vector<uint32_t> counts(2 * MV * S * NA * 2, 0);
    auto cnt = [&](int w, int s, int al) { return &counts[((static_cast<size_t>(w * MV) * S + s) * NA + al) * 2]; };
    cnt(0, 0, 0)[0] = 20; cnt(0, 0, 0)[1] = 20; cnt(0, 1, 0)[0] = 30; cnt(0, 1, 0)[1] = 30;   // 100 REF reads, no ALT
    cnt(1, 0, 0)[0] = 5; cnt(1, 0, 0)[1] = 5; cnt(1, 1, 0)[0] = 6; cnt(1, 1, 0)[1] = 6;       // 22 REF ...
    cnt(1, 1, 1)[0] = 3; cnt(1, 1, 1)[1] = 2;                                                 // ... and 5 ALT in the tumour
    ma_var_out_t v{};
    v.allele_pool = pool.data(); v.win_nvars = win_nvars.data(); v.var_pos = var_pos.data(); v.var_ref_off = var_ref_off.data();
    v.var_ref_len = var_ref_len.data(); v.var_nalts = var_nalts.data(); v.alt_off = alt_off.data(); v.alt_len = alt_len.data();
    v.alt_type = alt_type.data(); v.alt_length = alt_length.data();
    ma_geno_out_t q{};
    q.var_qual = var_qual.data(); q.allele_counts = counts.data();
    auto const all = RecordsOfBatch(p, fb, v, q, nullptr, /*supported_only=*/false);
    CHECK(all.size() == 2 && all[0].TotalCoverage() == 100 && all[1].TotalCoverage() == 27 && !all[0].HasAltSupport());
    auto const calls = RecordsOfBatch(p, fb, v, q);
    CHECK(calls.size() == 1 && calls[0].window_index == 1 && calls[0].pos1 == 900 && calls[0].HasAltSupport());
    VariantStore st;
    st.AddVariants(calls);
    auto const out = st.ExtractAll();
    CHECK(out.size() == 1 && out[0].qual == 33.0 && out[0].ad[1][1] == 5);
    VariantStore wrong;  // the order this test guards against: the site is lost
    wrong.AddVariants(all);
    CHECK(wrong.ExtractAll().empty());
    // flush-time rule (variant_store.cpp:62-66): ALT depth but every category REF -> not written
    VariantRecord refonly{0, 10, "A", {"A"}, 1.0, {{3, 2}, {1, 1}}, 0};
    refonly.alt_type = {-1};
    CHECK(refonly.HasAltSupport() && refonly.HasNoSupport());
  }
  {  // VCF text (caller/variant_call.cpp, caller/sample_format_data.cpp:32-98)
    // GL index -> genotype: 0/0 0/1 1/1 0/2 1/2 2/2 0/3 ... (variant_call.cpp:262-290)
    const int want[10][2] = {{0, 0}, {0, 1}, {1, 1}, {0, 2}, {1, 2}, {2, 2}, {0, 3}, {1, 3}, {2, 3}, {3, 3}};
    for (size_t g = 0; g < 10; ++g) {
      auto const gt = GenotypeOfPlIndex(g);
      CHECK(gt.first == want[g][0] && gt.second == want[g][1]);
    }
    // tests/caller/variant_call_test.cpp: AD 30,20 renders PRAD 5.51 / PANG 0.588 with set values; the functions themselves:
    CHECK(std::abs(PolarRadius(30, 20) - std::log10(1.0 + std::sqrt(1300.0))) < 1e-12);
    CHECK(std::abs(PolarAngle(20, 30) - std::atan2(20.0, 30.0)) < 0.005);  // the reference's minimax atan2, restated as is
    CHECK(std::abs(PolarAngle(20, 30) - 0.5906285633997478) < 1e-12);       // (its own arithmetic, evaluated independently)
    CHECK(std::abs(PolarAngle(0, 30)) < 0.005 && std::abs(PolarAngle(30, 0) - 1.5707963) < 0.005);
    CHECK(FormatComplexityScore(0.5) == "0.5" && FormatComplexityScore(2.0) == "2" && FormatComplexityScore(1.23456) == "1.235");
    VariantRecord r{0, 1500, "A", {"T", "G"}, 12.3456, {{12, 0, 0}, {9, 6, 1}}, 2};
    r.alt_type = {0, 0};
    r.alt_length = {1, 1};
    r.adf = {{6, 0, 0}, {5, 3, 1}};
    r.adr = {{6, 0, 0}, {4, 3, 0}};
    r.pl = {{0, 30, 300, 30, 300, 300}, {120, 0, 200, 90, 210, 400}};
    r.gq = {30, 90};
    r.sample_window_cov = {30.0, 32.0};
    std::string const line = AsVcfRecord(r, ref, {Tag::CTRL, Tag::CASE}, true);
    CHECK(line.rfind("chr1\t1500\t.\tA\tT,G\t12.35\t.\tCASE;MULTIALLELIC;TYPE=SNV,SNV;LENGTH=1,1\t", 0) == 0);
    CHECK(line.find(std::string("\t") + kVcfFormatKey + "\t0/0:12,0,0:6,0,0:6,0,0:12:") != std::string::npos);
    // SB (variant_support.cpp:197-237): ln((5+1)(3+0+1) / ((4+1)(3+1+1))) = ln(24/25) = -0.041; sample 0: ln(7*1/(7*1)) = 0
    CHECK(line.find("\t0/1:9,6,1:5,3,1:4,3,0:16:.:.:-0.041:.:.:.:.:.:.:0.50:") != std::string::npos);
    CHECK(line.find(":12:.:.:0.000:.:") != std::string::npos);
    CHECK(line.find(":120,0,200,90,210,400:90") != std::string::npos);
    size_t colons = 0;  // 24 FORMAT values per sample
    for (char c : line.substr(line.rfind('\t'))) colons += c == ':';
    CHECK(colons == 23);
    VariantRecord none = r;
    none.ad = {{0, 0, 0}, {9, 6, 1}};
    CHECK(AsVcfRecord(none, ref, {Tag::CTRL, Tag::CASE}, true).find("\t./.:.:.:") != std::string::npos);
    std::vector<SampleInfo> ss{{"normal", Tag::CTRL, nullptr, 0, 0, 0}, {"tumor", Tag::CASE, nullptr, 1, 0, 0}};
    std::string const hdr = VcfHeader(ref, ss, true, true, "cmd", "ref.fa");
    CHECK(hdr.rfind("##fileformat=VCFv4.5\n", 0) == 0 && hdr.find("##contig=<ID=chr1,length=10000>") != std::string::npos);
    CHECK(hdr.find("##FORMAT=<ID=PL,Number=G,Type=Integer") != std::string::npos && hdr.find("##INFO=<ID=SEQ_CX,Number=11") != std::string::npos);
    CHECK(hdr.find("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tnormal\ttumor\n") != std::string::npos);
  }
  std::printf("host units ok\n");
  return 0;
}
