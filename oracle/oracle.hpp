// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.hpp header).
// C++ data model of the CPU restatement.  The flat C entry points that tests drive
// through ctypes live in capi.cpp.
#pragma once
#include "common.hpp"

namespace orc {

// cbdg/label.h:8-37
enum LabelTag : u8 { L_REFERENCE = 1, L_CTRL = 2, L_CASE = 4 };

// cbdg/read.h:19-118 -- only the fields the path consumes.
struct Read {
  std::string_view seq;
  const u8* qual = nullptr;
  u32 qname_id = 0;   // host-interned QNAME (replaces the string in MateMer, graph.h:102-117)
  u8 sample = 0;      // SampleIndex()
  u8 role = 0;        // 0 = CTRL (normal), 1 = CASE (tumor): RoleIndex(), node.h:23-25
  bool pass = true;   // PassesAlnFilters(): mapq >= 20 (read.h:35-38)
  bool rev = false;   // SAM flag 0x10 (strand for evidence, genotyper.cpp:427)
  LabelTag Tag() const { return role == 1 ? L_CASE : L_CTRL; }
};

// cbdg/graph_params.h:29-53 (+ the constants hard-coded in graph.cpp / max_flow.h)
struct Params {
  u32 min_k = 13, max_k = 127, k_step = 6;
  u32 min_node_cov = 2, min_anchor_cov = 5;
  u32 num_samples = 2;
  u32 min_anchor_len = 150;     // graph.cpp:88
  u32 max_mismatch = 2;         // graph.h:129
  u32 bfs_limit = 1u << 20;     // max_flow.h:69
};

// cbdg/graph_complexity.h:155-161
struct GraphComplexity {
  u64 cyclomatic = 0, branch_points = 0, max_dir_degree = 0;
  f64 unitig_ratio = 0.0, coverage_cv = 0.0, tip_to_path = 0.0;
  bool IsComplex() const { return cyclomatic >= 50 && branch_points >= 50; }  // :112-121
};

// cbdg/path.h:19-85 -- assembled haplotype.
struct Haplotype {
  std::string seq;
  std::vector<std::pair<u32, u32>> node_weights;  // (Confidence, bases contributed) runs
  std::vector<u32> node_covs;                      // per-node TotalReadSupport, walk order
  f64 mean_cov = 0, median_cov = 0, sd_cov = 0, cv_cov = 0, qcv_cov = 0, total_cov = 0;
  u32 MinWeight() const;                           // path.cpp:34-37
  std::vector<u32> PerBaseWeights() const;         // path.cpp:25-32
  void Finalize();                                 // path.cpp:39-70
};

// cbdg/component_result.h:32-81
struct ComponentResult {
  std::vector<Haplotype> haps;  // [0] = REF
  GraphComplexity metrics;
  u32 anchor_start = 0;
  bool hit_bfs_limit = false;
  // MaxAltPathCv (component_result.cpp:50-58): -1 when there is no ALT haplotype.
  f64 MaxAltPathCv() const;
};

struct AssemblyResult {
  std::vector<ComponentResult> comps;
  u32 used_k = 0;  // Graph::CurrentK() after the call
  bool hit_bfs_limit = false;  // MaxFlow::HitTraversalLimit in some component of the reported attempt
};

extern unsigned long long g_debug_counters[4];  // graph.cpp: cycle / complexity gate / traversal limit events

// repeat.cpp
usize HammingDist(std::string_view a, std::string_view b);
bool HasRepeat(std::string_view seq, usize k, usize max_mm);

// graph.cpp -- Graph::BuildComponentResults (cbdg/graph.cpp:78-256)
AssemblyResult BuildComponentResults(std::string_view ref, const std::vector<Read>& reads,
                                     const Params& prm);

// --- haplotype <-> reference POA (caller/msa_builder.*; SPOA 4.1.5 restated) -------------
struct PoaNode {
  u32 id = 0;
  u8 code = 0;
  std::vector<u32> in_edges, out_edges;  // edge indices
  std::vector<u32> aligned;              // node ids
};
struct PoaEdge {
  u32 tail = 0, head = 0;
  i64 weight = 0;
  std::vector<u32> labels;
};
struct PoaGraph {
  std::vector<PoaNode> nodes;
  std::vector<PoaEdge> edges;
  std::vector<i32> seq_first;   // first node id of each sequence (-1: empty)
  std::vector<u32> rank_to_node;
  i32 coder[256];
  u8 decoder[256];
  u32 num_codes = 0;
  PoaGraph() { Clear(); }
  void Clear();
  i32 Successor(u32 node, u32 label) const;  // spoa::Graph::Node::Successor
};
struct PoaScoring {  // spoa::AlignmentEngine::Create(kNW, m, n, g, e, q, c)
  i32 m = 0, n = -6, g = -6, e = -2, q = -26, c = -1;  // caller/msa_builder.h:72-77
};
using PoaAlignment = std::vector<std::pair<i32, i32>>;  // (node id | -1, seq pos | -1)
// score_out (optional): the DP's optimum H(sink, L) the backtrack starts from
PoaAlignment PoaAlign(const PoaScoring& sc, std::string_view seq, const PoaGraph& g, i32* score_out = nullptr);
void PoaAddAlignment(PoaGraph& g, const PoaAlignment& aln, std::string_view seq,
                     const std::vector<u32>& weights);
// caller/msa_builder.cpp:29-42
void UpdateSpoaState(PoaGraph& g, const PoaScoring& sc, const std::vector<std::string_view>& seqs,
                     const std::vector<std::vector<u32>>& weights);

// --- POA DAG -> variants (caller/variant_extractor.cpp, variant_bubble.cpp, raw_variant.cpp)
enum AlleleType : i8 { T_REF = -1, T_SNV = 0, T_INS = 1, T_DEL = 2, T_MNP = 3, T_CPX = 4 };
struct AltAllele {
  std::string seq;
  std::vector<std::pair<u32, u32>> hap_starts;  // (hap idx, local start0 on that hap), hap asc
  i64 length = -1;
  AlleleType type = T_REF;
};
struct RawVariant {
  u64 pos1 = 0;         // mGenomeChromPos1
  u64 ref_start0 = 0;   // mLocalRefStart0Idx
  std::string ref;
  std::vector<AltAllele> alts;  // sorted by sequence
};
std::vector<RawVariant> ExtractVariants(const PoaGraph& g, u64 ref_anchor_pos1);

// --- read <-> haplotype aligner (caller/genotyper.cpp:376-411; minimap2 2.30 replaced by the
//     canonical seed-anchored overlap DP documented in DESIGN.md -- scores pinned to a brute force,
//     parity against minimap2 itself UNPINNED) ---
struct CigarUnit { char op; u32 len; };
struct AlnResult {
  bool hit = false;
  i32 score = 0, rs = 0, re = 0, qs = 0, qe = 0;
  u32 hap = 0;
  std::vector<CigarUnit> cigar;  // incl. leading/trailing S (genotyper.cpp:45-69)
};
struct AlignParams { i32 seed_k = 11; i32 min_score = 80; };
i32 AlignReach(i32 m, i32 min_score);  // K of the search region R = [vmin - K, vmax + K]
bool SeedDiagonals(const std::vector<u8>& q, const std::vector<u8>& t, i32 seed_k, i32 K, i32* vmin, i32* vmax);
AlnResult AlignReadToHap(std::string_view read, std::string_view hap, const AlignParams& ap);

// --- scoring epilogue + evidence (local_scorer.cpp, combined_scorer.cpp, genotyper.cpp:269-456)
struct Assignment {
  bool valid = false;
  f64 local_score = 0, local_identity = 0, folded_pos = 0;
  i32 global_score = 0;
  u32 ref_nm = 0, own_nm = 0, hap_id = 0;
  u32 allele = 0;
  u8 base_qual = 0;
  f64 Combined() const { return static_cast<f64>(global_score) + local_score * local_identity; }
};
// per read: one Assignment per variant (valid=false when no overlapping alignment)
std::vector<Assignment> AssignReadToAlleles(const Read& rd, const std::vector<std::string>& haps,
                                            const std::vector<RawVariant>& vars,
                                            const AlignParams& ap,
                                            std::vector<AlnResult>* alns_out = nullptr);

// caller/genotype_likelihood.cpp:93-307: Dirichlet-multinomial genotype PLs (VCF genotype order) and GQ
std::vector<u32> ComputeGenotypePLs(const std::vector<int>& allele_counts);
u32 ComputeGenotypeQuality(const std::vector<u32>& pls);

// hts/cigar_utils.h:48-139
u32 ComputeEditDistance(const std::vector<CigarUnit>& cigar, const std::vector<u8>& q,
                        const u8* t, usize tlen);
usize CigarRefPosToQueryPos(const std::vector<CigarUnit>& cigar, usize ref_pos);

// --- sequence-complexity annotation (seqcx.cpp) -------------------------------
struct LongdustQ {  // base/longdust_scorer.h:217-462
  std::vector<f64> F;
  f64 gc;
  int k;
  u32 mask, num_kmers;
  explicit LongdustQ(int kmer_len = 7, int max_len = 1024, f64 gc_frac = 0.41);
  f64 ComputeF(int ell) const;
  f64 ScoreOneStrand(std::string_view seq) const;
  f64 Score(std::string_view seq) const;
};
struct TandemRepeat {  // base/sequence_complexity.h:28-47
  i32 period = 0;
  f32 copies = 0.0F;
  i32 start = 0, span = 0, errors = 0;
  bool exact = false;
  f32 Purity() const { return span <= 0 ? 0.0F : 1.0F - (static_cast<f32>(errors) / static_cast<f32>(span)); }
};
struct TrFeatures { i32 dist = -1, period = 0; f32 purity = 0.0F; i32 stutter = 0; };  // sequence_complexity.h:52-58
struct HapRegion { std::string_view hap; usize pos = 0, len = 0; };                    // sequence_complexity.h:16-21
struct SeqCx {  // base/sequence_complexity.h:106-158 (11 features, VCF SEQ_CX order)
  i32 ctx_hrun = 0;
  f32 ctx_entropy = 0.0F;
  f64 ctx_flank_lq = 0.0, ctx_hap_lq = 0.0;
  i32 delta_hrun = 0;
  f32 delta_entropy = 0.0F;
  f64 delta_flank_lq = 0.0;
  f32 tr_affinity = 0.0F, tr_purity = 0.0F;
  i32 tr_period = 0, stutter = 0;
  void MergeMax(const SeqCx& o);
};
std::string_view ExtractFlank(std::string_view hap, usize pos, usize len, i64 flank);
i32 MaxHomopolymerRun(std::string_view s);
f32 LocalShannonEntropy(std::string_view s);
std::vector<TandemRepeat> FindExactRepeats(std::string_view s, i32 max_period = 6, f32 min_copies = 2.5F);
std::vector<TandemRepeat> FindApproxRepeats(std::string_view s, i32 max_period = 6, f32 min_copies = 3.0F,
                                            i32 max_edits = 1);
TrFeatures FlattenTRFeatures(const std::vector<TandemRepeat>& rs, i32 vpos, i32 vlen);
struct SeqCxScorer {  // base/sequence_complexity.h:186-300
  LongdustQ flank, hap;
  explicit SeqCxScorer(f64 gc_frac = 0.41);
  SeqCx Score(const HapRegion& ref, const HapRegion& alt) const;
};
struct AltSites { usize len = 0; std::vector<std::pair<u32, u32>> hap_starts; };
// core/variant_annotator.cpp:43-85 for one variant; haps[0] is the component's REF haplotype
SeqCx AnnotateVariant(const SeqCxScorer& sc, const std::vector<std::string_view>& haps, usize ref_pos, usize ref_len,
                      const std::vector<AltSites>& alts);

}  // namespace orc
