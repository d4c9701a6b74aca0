/* microasm.h -- C-ABI of the MI355X-native per-window microassembly engine.
 *
 * Drop-in boundary for Lancet2's per-window worker (core/variant_builder.cpp:201-276).  The
 * reference has no FFI layer; each entry point below replaces one C++ seam inside
 * VariantBuilder::ProcessWindow and is batched over windows (the unit of data parallelism):
 *
 *   ma_repeat_gate_batch   <- base::HasRepeat / HasExactRepeat            (base/repeat.h:22-27;
 *                             callers cbdg/graph.h:127-131, core/variant_builder.cpp:116-117)
 *   ma_assemble_batch      <- cbdg::Graph::BuildComponentResults          (cbdg/graph.h:53)
 *   ma_msa_batch           <- caller::MsaBuilder::UpdateSpoaState + caller::VariantSet ctor
 *                             (caller/msa_builder.h:81-92, caller/variant_set.h:25)
 *   ma_genotype_batch      <- caller::Genotyper::Genotype                 (caller/genotyper.h:219)
 *   ma_annotate_batch      <- core::VariantAnnotator::Annotate{Sequence,Graph}Complexity (core/variant_annotator.h)
 *   ma_process_batch       <- the chained body of ProcessWindow           (core/variant_builder.cpp:229-262)
 *
 * Conventions: plain pointers and sizes only; all buffers are caller owned, struct-of-arrays;
 * every function returns 0 on success and a negative ma_error otherwise and never throws.
 * Pointers are HOST pointers when the context was created with MA_MEM_HOST (the library stages
 * them through HBM itself) and DEVICE pointers with MA_MEM_DEVICE (zero-copy; the bench path).
 * Reads of a window must arrive in the collector's sorted order (core/read_collector.cpp:42-53).
 */
#ifndef MICROASM_H_
#define MICROASM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MA_VERSION 2

enum ma_error {
  MA_OK = 0,
  MA_ERR_ARG = -1,      /* bad argument / inconsistent sizes */
  MA_ERR_NO_DEVICE = -2,/* no HIP device: the engine has NO CPU fallback */
  MA_ERR_HIP = -3,      /* HIP runtime error (see ma_last_error) */
  MA_ERR_NOMEM = -4,    /* workspace allocation failed */
  MA_ERR_PARAM = -5     /* parameter out of the supported range */
};

enum ma_memspace { MA_MEM_HOST = 0, MA_MEM_DEVICE = 1 };

/* Per-window status bits (0 == assembled cleanly). Mirrors the skip reasons of
 * VariantBuilder::StatusCode (core/variant_builder.h:73-83) plus engine capacity flags. */
enum ma_wstatus {
  MA_W_NO_HAPLOTYPE = 1u << 0,   /* SKIPPED_NOASM_HAPLOTYPE: no ALT haplotype at any k */
  MA_W_HAP_OVERFLOW = 1u << 1,   /* more haplotypes / components than max_haps / max_comps */
  MA_W_LEN_OVERFLOW = 1u << 2,   /* haplotype longer than max_hap_len or more runs than max_runs */
  MA_W_BFS_LIMIT = 1u << 3,      /* MaxFlow::HitTraversalLimit (max_flow.h:69) in some component */
  MA_W_TABLE_OVERFLOW = 1u << 4, /* k-mer table / edge list / search arena capacity exceeded, or the traversal cap fell where
                                    the folded walk search cannot place it: the window's result is not the reference's */
  MA_W_VAR_OVERFLOW = 1u << 5,   /* more variants / alleles / allele bytes than the caps */
  MA_W_CIGAR_OVERFLOW = 1u << 6, /* a read<->haplotype CIGAR had more than max_cigar operations and was cut before the scoring
                                    epilogue (local_scorer.cpp:166-279 scores the whole CIGAR): re-submit the window with a
                                    larger max_cigar -- 2 * ((read length - 80) / 15) + 3 operations always suffice */
  MA_W_READ_OVERFLOW = 1u << 7   /* the window holds a read longer than the stage supports (assembly: 1024 bases, genotyping:
                                    608): the stage skips the window (no haplotypes / no allele counts) instead of failing
                                    the batch */
};

/* GraphParams (cbdg/graph_params.h:29-53) + the constants the reference hard-codes + engine caps. */
typedef struct ma_params {
  int32_t min_k, max_k, k_step;          /* 13, 127, 6 */
  int32_t min_node_cov, min_anchor_cov;  /* 2, 5 */
  int32_t num_samples;                   /* GraphParams::mNumSamples */
  int32_t min_anchor_len;                /* 150  (graph.cpp:88) */
  int32_t max_mismatch;                  /* 2    (graph.h:129) */
  int32_t bfs_limit;                     /* 1<<20 (max_flow.h:69) */
  /* read<->haplotype aligner (genotyper.cpp:89-191 as restated in DESIGN.md).  There is no band parameter: like the
   * reference (bw = 10000, genotyper.cpp:140) the search region never excludes an alignment its seeds support; it is
   * derived per pair from the seed diagonals, the read length and min_aln_score. */
  int32_t aln_tier;                      /* testing knob, results NEVER depend on it.  0 (default): every pair takes the
                                            cheapest exact route.  bit 0: all DP pairs through the generic any-width kernel;
                                            bit 1: the gapless certificates are off (every seeded pair runs the DP) */
  int32_t min_aln_score;                 /* 80 (minimap2 min_dp_max default) */
  /* engine caps (outputs are fixed-stride; overflow sets a status bit) */
  int32_t max_comps;                     /* components kept per window (default 4) */
  int32_t max_haps;                      /* haplotype slots per window, REF included (default 16, at most 32: a component of
                                            17 .. 32 haplotypes takes the POA's 32-bit-label kernels; the reference has no cap,
                                            cbdg/graph.cpp:846-924 -- beyond 32 the window is flagged MA_W_HAP_OVERFLOW) */
  int32_t max_hap_len;                   /* bytes per haplotype slot (default 2048) */
  int32_t max_runs;                      /* (weight,nbases) runs per haplotype (default 256) */
  int32_t max_vars;                      /* variants per window (default 64) */
  int32_t max_alts;                      /* ALT alleles per variant (default 4) */
  int32_t max_allele_bytes;              /* allele-string pool per window (default 4096) */
  int32_t max_cigar;                     /* CIGAR ops kept per read x haplotype alignment (default 16) */
  int32_t case_ctrl_mode;                /* 1: QUAL = SOLOR (variant_call.cpp:316-345) */
} ma_params_t;

void ma_default_params(ma_params_t* p);

/* One batch of windows.  Window w owns reference bytes [ref_off[w], ref_off[w+1]) and reads
 * [read_win_off[w], read_win_off[w+1]); read r owns bases/quals [read_off[r], read_off[r+1]).
 * Limits: windows up to 8192 bases (repeat gate), reads up to 608 bases (ma_genotype_batch answers MA_ERR_PARAM
 * beyond: short-read data only, like the reference's 150 bp workloads). */
typedef struct ma_batch {
  int32_t n_windows;
  int64_t n_reads;
  const uint8_t* ref_bases;       /* ASCII, already normalised to ACGTN (hts/reference.cpp:176-194) */
  const uint32_t* ref_off;        /* [n_windows + 1] */
  const uint32_t* read_win_off;   /* [n_windows + 1] */
  const uint64_t* read_off;       /* [n_reads + 1] */
  const uint8_t* read_bases;      /* ASCII (hts/alignment.cpp:123-143 decode) */
  const uint8_t* read_quals;      /* Phred */
  const uint32_t* read_qname_id;  /* host-interned QNAME, unique per distinct name within a window */
  const uint8_t* read_sample;     /* cbdg::Read::SampleIndex() */
  const uint8_t* read_flags;      /* MA_RF_* */
  /* OPTIONAL (may be NULL) performance hint: window-relative 0-based reference offset at which base 0 of
   * the read is expected to align (cbdg::Read::StartPos0() - window start - leading soft clip), or
   * MA_NO_HINT.  Results never depend on it: k-mers that equal the reference k-mer at the hinted offset
   * skip the hash table, everything else takes the general path. */
  const int32_t* read_hint;
} ma_batch_t;

#define MA_NO_HINT INT32_MIN

enum ma_read_flags {
  MA_RF_PASS = 1u << 0,  /* cbdg::Read::PassesAlnFilters(): mapq >= 20 (cbdg/read.h:35-38) */
  MA_RF_CASE = 1u << 1,  /* Label::CASE (tumor) if set, Label::CTRL (normal) otherwise */
  MA_RF_REV = 1u << 2    /* SAM flag 0x10: reverse strand (genotyper.cpp:427) */
};

/* ---- repeat gate ------------------------------------------------------------------------ */
/* max_approx[w] = length of the longest pair of equal-length substrings at different offsets of
 * window w that differ in <= max_mismatch positions; max_exact[w] likewise with 0 mismatches.
 * HasRepeat(SlidingView(ref,k), mm) (base/repeat.cpp:348-371) == (max_approx[w] >= k);
 * HasExactRepeat(SlidingView(ref,max_k)) (variant_builder.cpp:116-117) == (max_exact[w] >= max_k). */
typedef struct ma_gate_out {
  uint32_t* max_approx; /* [n_windows] */
  uint32_t* max_exact;  /* [n_windows] */
} ma_gate_out_t;

/* ---- assembly output (ComponentResult, cbdg/component_result.h:32-81) -------------------- */
typedef struct ma_asm_out {
  uint32_t* win_status;   /* [n]            ma_wstatus bits */
  uint32_t* win_k;        /* [n]            Graph::CurrentK() */
  uint32_t* win_ncomp;    /* [n]            components with >= 1 walk */
  /* per component, stride max_comps */
  uint32_t* comp_anchor;  /* AnchorStartOffset() */
  uint32_t* comp_hap0;    /* first haplotype slot of this component (REF) */
  uint32_t* comp_nhaps;   /* NumPaths() */
  uint32_t* comp_cx;      /* [.. * 3] cyclomatic, branch points, max single-direction degree */
  double* comp_cxf;       /* [.. * 4] unitig ratio, coverage CV, tip/path cov ratio, MaxAltPathCv (-1: none) */
  /* per haplotype slot, stride max_haps */
  uint32_t* hap_len;      /* Path::Sequence().size() */
  uint32_t* hap_nruns;    /* number of (weight, nbases) runs (Path::mNodeWeights) */
  double* hap_stats;      /* [.. * 6] mean, median, sd, cv, qcv, total coverage (path.cpp:39-70) */
  uint8_t* hap_bases;     /* [.. * max_hap_len] */
  uint32_t* hap_runs;     /* [.. * max_runs * 2] weight, nbases */
} ma_asm_out_t;

/* ---- MSA + variant extraction (RawVariant, caller/raw_variant.h) ------------------------- */
typedef struct ma_var_out {
  uint32_t* win_nvars;    /* [n] */
  /* per variant, stride max_vars */
  uint32_t* var_comp;     /* component index within the window */
  uint32_t* var_pos;      /* 0-based offset in the window of mGenomeChromPos1 (pos1 = StartPos1 + var_pos) */
  uint32_t* var_ref_start;/* mLocalRefStart0Idx */
  uint32_t* var_ref_off;  /* REF allele: offset/len into this window's allele pool */
  uint32_t* var_ref_len;
  uint32_t* var_nalts;
  /* per (variant, alt), stride max_alts */
  uint32_t* alt_off;
  uint32_t* alt_len;
  int32_t* alt_type;      /* AlleleType: 0 SNV 1 INS 2 DEL 3 MNP 4 CPX (caller/alt_allele.h:14) */
  int32_t* alt_length;    /* AltAllele::mLength */
  /* per (variant, haplotype-of-its-component), stride max_haps: which allele the haplotype carries
   * (0 = REF, a+1 = ALT a) and where the bubble starts on it (AltAllele::mLocalHapStart0Idxs) */
  uint8_t* var_hap_allele;
  uint32_t* var_hap_start;
  uint8_t* allele_pool;   /* [n * max_allele_bytes] */
} ma_var_out_t;

/* ---- genotyping (VariantSupport, caller/variant_support.cpp:24-66) ----------------------- */
typedef struct ma_geno_out {
  /* [n * max_vars * num_samples * (max_alts+1) * 2]: fwd, rev read counts per allele (AD = fwd+rev) */
  uint32_t* allele_counts;
  double* var_qual;       /* [n * max_vars] site QUAL: SOLOR in case/ctrl mode (variant_call.cpp:316-345), else the largest
                           * PL[0/0] over the samples with evidence (variant_call.cpp:289-303) */
  /* optional debug taps (may be NULL): per read x haplotype slot of the read's window */
  int32_t* aln_rec;       /* [n_reads * max_haps * 6] hit, score, rs, re, qs, qe */
  uint32_t* aln_cigar;    /* [n_reads * max_haps * (1 + max_cigar)] n_ops, then len<<4|op (op: 0 M 1 I 2 D 4 S) */
  /* optional: per read x variant: allele (255 = none) and CombinedScore */
  uint8_t* asg_allele;    /* [n_reads * max_vars] */
  double* asg_score;      /* [n_reads * max_vars] */
  /* optional (may be NULL): FORMAT PL and GQ of every sample with evidence -- Dirichlet-multinomial genotype
   * likelihoods over the sample's allele depths (caller/genotype_likelihood.cpp:93-272), VCF genotype order
   * (0/0, 0/1, 1/1, 0/2, ...), G = (max_alts + 1)(max_alts + 2) / 2 slots of which the first K(K+1)/2 are used for a
   * variant with K alleles; zero for samples without evidence */
  uint32_t* var_pl;       /* [n * max_vars * num_samples * G] */
  uint32_t* var_gq;       /* [n * max_vars * num_samples] */
} ma_geno_out_t;

/* ---- variant annotation (core/variant_annotator.cpp:43-101; VCF INFO SEQ_CX / GRAPH_CX) ------ */
typedef struct ma_cx_out {
  /* per variant, stride max_vars.  SequenceComplexity (base/sequence_complexity.h:106-158), merged over the
   * variant's (ALT, haplotype) sites with MergeMax exactly as AnnotateSequenceComplexity does */
  int32_t* seq_cx_i;  /* [.. * 4] ContextHRun, DeltaHRun, TrPeriod, IsStutterIndel */
  float* seq_cx_f;    /* [.. * 4] ContextEntropy, DeltaEntropy, TrAffinity, TrPurity */
  double* seq_cx_d;   /* [.. * 3] ContextFlankLQ, ContextHaplotypeLQ, DeltaFlankLQ */
  double* graph_cx;   /* [.. * 3] GraphEntanglementIndex, TipToPathCovRatio, MaxSingleDirDegree
                       * (caller::GraphMetrics, variant_annotator.cpp:87-99) */
} ma_cx_out_t;

typedef struct ma_ctx ma_ctx_t;

int ma_create(const ma_params_t* prm, int device, int memspace, ma_ctx_t** out);
void ma_destroy(ma_ctx_t* ctx);
const char* ma_last_error(const ma_ctx_t* ctx);
/* HIP stream all launches and copies go to (void* == hipStream_t).  A context starts on a non-blocking stream of its
 * own, so that several contexts on one device overlap; NULL selects the legacy null stream. */
int ma_set_stream(ma_ctx_t* ctx, void* hip_stream);
int ma_synchronize(ma_ctx_t* ctx);

int ma_repeat_gate_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_gate_out_t* out);
int ma_assemble_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_asm_out_t* out);
int ma_msa_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_asm_out_t* asmb, const ma_var_out_t* out);
int ma_genotype_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_asm_out_t* asmb,
                      const ma_var_out_t* vars, const ma_geno_out_t* out);
/* VariantAnnotator::AnnotateSequenceComplexity + AnnotateGraphComplexity (core/variant_annotator.cpp:43-101,
 * called from VariantBuilder::ExtractVariants, core/variant_builder.cpp:159-160) for every variant of the batch.
 * gc_frac = the --genome-gc-bias background GC fraction of the LongdustQ null model (default 0.41).
 * Integer features are exact; the f32/f64 features go through device log2f/log1p/log10. */
int ma_annotate_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_asm_out_t* asmb, const ma_var_out_t* vars,
                      double gc_frac, const ma_cx_out_t* out);
/* gate -> assemble -> msa -> genotype on one batch; any output struct may be NULL-filled only in
 * its optional members. */
int ma_process_batch(ma_ctx_t* ctx, const ma_batch_t* b, const ma_gate_out_t* gate,
                     const ma_asm_out_t* asmb, const ma_var_out_t* vars, const ma_geno_out_t* geno);

/* MA_MEM_HOST only (a no-op otherwise): hand over the batch the caller will pass to ma_process_batch NEXT and return at
 * once -- the staged pipeline of the reference's AsyncWorker (core/async_worker.cpp:47-110: extract window j + 1 while window
 * j is assembled) at batch granularity:
 *     ma_prefetch_batch(ctx, &batch[j + 1]);  ma_process_batch(ctx, &batch[j], ...);
 * The context uploads the batch in the background (an uploader thread, 16 MB pieces) and queues its COMPUTE behind whatever
 * its lanes are doing: a lane goes from the last kernel of batch j straight to the first of batch j + 1, and
 * ma_process_batch(j + 1) only waits for the packed records and scatters them into the caller's arrays.  Work that was
 * queued ahead assumes that the call will ask for the same optional output arrays as the previous ma_process_batch did, with
 * the same parameters and timing mode; if it does not (or asks for the per-read debug taps), the queued results are dropped
 * and the batch is computed in the call -- results never depend on whether, or how, a batch was prefetched.
 * `next` and the arrays it points to must stay unchanged until the ma_process_batch call that consumes it (recognised by
 * the struct's address, window and read counts) has returned; the arrays should be page-locked (hipHostMalloc /
 * hipHostRegister), a copy from pageable memory is slow.  At most two batches wait at a time; further calls do nothing. */
int ma_prefetch_batch(ma_ctx_t* ctx, const ma_batch_t* next);

/* Kernel timing mode: 0 = off, 1 = reset at every API call (default), 2 = accumulate across calls
 * until ma_timing_control is called again (used by bench.py to time kernels over the timed region), 3 = as 2 and
 * one extra small kernel per k attempt gathers the workload statistics ma_last_stats reports in out[8..13]
 * (bench.py runs one untimed step in this mode). */
int ma_timing_control(ma_ctx_t* ctx, int mode);

/* Per-kernel timing of the last call, measured with HIP events on the launch stream.
 * names[i] points to a static string; returns the number of entries written (<= cap). */
int ma_last_kernel_times(ma_ctx_t* ctx, const char** names, float* ms, int cap);

/* ma_process_batch runs a batch as `n` contiguous window ranges concurrently, each on its own HIP stream and
 * workspace (the stages have complementary bottlenecks, so two ranges in flight fill each other's gaps).
 * n = 0 (default): automatic -- 3 for batches of >= 6144 windows, 2 for >= 2048, else 1.  Results do not depend on n.
 * The caller's stream (ma_set_stream) still orders the call as a whole. */
int ma_set_streams(ma_ctx_t* ctx, int n);

/* Work counters accumulated over the same region as ma_last_kernel_times (reset by ma_timing_control):
 *   out[0] read x haplotype pairs seen by ma_genotype_batch      out[1] pairs that needed the DP
 *   out[2] windows passed to ma_assemble_batch                   out[3] k attempts x windows assembled
 *   out[4..7] DP pairs by region width: <= 41, <= 65, <= 129 diagonals, wider (any-width kernel)
 *   out[8..13] (timing mode 3 only) distinct k-mers, nodes after the first low-coverage pass, k-mer instances on the
 *              hash-table path, k-mer instances, (k+1)-mers of reads queued for the edge builder, read-support counts
 *              queued -- each summed over the window attempts
 *   out[14..19] (timing mode 3 only) DP cells: [14] cells of the read aligner's DP regions (rows x region width, summed over
 *              the pairs that ran the DP), [15] cells of the POA's banded fills (rows x band columns), [16] banded fills,
 *              [17] cells of the POA's full fills (rows x haplotype length), [18] haplotype <-> graph alignments,
 *              [19] alignments written down in closed form (no fill) -- what bench.py's `gcups` block divides the kernel times by
 * Used by bench.py to price the kernels' algorithmic HBM bytes.  Returns the number of entries written. */
int ma_last_stats(ma_ctx_t* ctx, unsigned long long* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* MICROASM_H_ */
