/* hhviterbi_hip.h -- C ABI of the MI355X Viterbi HMM-HMM alignment engine (libhhviterbi_hip.so).
 *
 * Drop-in boundary for the batched Viterbi stage of HH-suite.  Each entry point replaces one piece
 * of the reference's C++ interface (file:line in soedinglab/hh-suite v3.3.0):
 *
 *   hhv_create / hhv_destroy   Viterbi::Viterbi(maxres, local, egq, egt, corr, min_overlap, shift,
 *                              ss_mode, ssw, S73, S33, S37)            src/hhviterbi.h:53-56,
 *                              constructed per thread in ViterbiConsumerThread  src/hhviterbirunner.h:21-34
 *   hhv_set_query              HMMSimd::MapOneHMM(q)                   src/hhhmmsimd.cpp:73-79,
 *                              called from HHblits::run                src/hhblits.cpp:1136
 *   hhv_upload_templates       HMMSimd::MapHMMVector(templates)        src/hhhmmsimd.cpp:86-160,
 *                              called per batch from ViterbiRunner::alignment  src/hhviterbirunner.cpp:151
 *   hhv_align                  Viterbi::Align(q, t, matrix, n, ss_mode) -> ViterbiResult{i[],j[],score[]}
 *                                                                      src/hhviterbi.cpp:163-191, src/hhviterbi.h:21-32
 *   hhv_set_celloff            ViterbiMatrix::setCellOff(i,j,elem,true) / Viterbi::ExcludeAlignment
 *                                                                      src/hhviterbimatrix-inl.h:28-35, src/hhviterbi.cpp:61-77
 *   hhv_backtrace              Viterbi::Backtrace(matrix, elem, i[], j[]) -> BacktraceResult of one template
 *                                                                      src/hhviterbi.cpp:83-160, src/hhviterbi.h:34-40
 *   hhv_hits / hhv_hit_path    the same walk for ALL templates of a set on the device, followed by
 *                              Viterbi::ScoreForBacktrace(...) -> BacktraceScore and the Hit fields filled in
 *                              ViterbiConsumerThread::align            src/hhviterbi.cpp:195-281, src/hhviterbirunner.cpp:35-62
 *   hhv_set_ss_tables / hhv_set_query_ss / hhv_upload_templates_ss / hhv_set_ss_mode
 *                              the secondary-structure inputs of Viterbi::Align...AndSS: S73/S33/S37 of the
 *                              Viterbi constructor, HMM::ss_pred/ss_conf/ss_dssp (src/hhhmm.h:151-154), and the
 *                              ss_hmm_mode argument of Viterbi::Align (src/hhviterbi.cpp:163-177)
 *   hhv_topk                   (new) device-side selection of the K best hits by Hit.score, the
 *                              per-GPU half of the sharded top-K merge
 *   hhv_tset_set_global_ids / hhv_merge_hits
 *                              (new) the other half: global template ids in the selected records and the merge of the
 *                              gathered lists into one hit list ordered like the reference's (src/hhhit.h:116-126)
 *
 * Conventions
 *   - plain C: pointers and sizes only, no C++/torch types.  Host pointers unless a parameter is
 *     documented as a DEVICE pointer.
 *   - every function returns HHV_OK (0) or a negative hhv_status; the library never calls exit()
 *     (the reference exits with the codes of src/hhsearch.h:6); hhv_last_error() gives the text.
 *   - "prepared profiles" = what Viterbi::Align sees after PrepareTemplateHMM
 *     (src/hhfunc.cpp:165-202):  p[(L+1)*20] row 0 unused,  tr[(L+1)*7] log2 scores in the enum
 *     order M2M,M2I,M2D,I2M,I2I,D2M,D2D (src/hhdecl.h:68).
 *   - one hhv_ctx is used from one host thread at a time; several contexts may coexist (one per
 *     GPU / per OpenMP thread, like the per-thread Viterbi objects of the reference).
 *   - there is no CPU fallback: without a usable HIP device hhv_create fails with HHV_E_DEVICE.
 */
#ifndef HHVITERBI_HIP_H
#define HHVITERBI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what this header declares is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define HHV_ABI_VERSION 1

typedef enum {
  HHV_OK = 0,
  HHV_E_ARG = -1,     /* bad argument (reference: exit(4)/(6)) */
  HHV_E_DEVICE = -2,  /* no HIP device / HIP runtime error / a kernel reported a failure in the context's device error word.
                       * Device-side failures (a wave that waited in vain for its partner, an illegal backtrace state) surface ONLY
                       * through the calls that wait for the stream themselves - hhv_sync, hhv_align, hhv_hits, hhv_topk,
                       * hhv_merge_hits, hhv_hit_path, hhv_backtrace_matrix, hhv_mac_realign* - : a caller that waits on
                       * hhv_stream() or an event of its own (hhv_align_async with device results) calls hhv_check_error after it. */
  HHV_E_MEMORY = -3,  /* host or device allocation failed (reference: exit(3)) */
  HHV_E_STATE = -4,   /* call order (no query set, no backtrace computed, ...) */
  HHV_E_LIMIT = -5    /* size beyond what this build supports */
} hhv_status;

/* Viterbi::Viterbi arguments that the hot path consumes (src/hhviterbirunner.h:32-33) */
typedef struct {
  int32_t device;   /* HIP device ordinal */
  int32_t local;    /* par.loc: 1 = local (Smith-Waterman like), 0 = global */
  float egq;        /* par.egq  end-gap penalty query */
  float egt;        /* par.egt  end-gap penalty template */
  float shift;      /* par.shift */
  float corr;       /* par.corr */
  float ssw;        /* par.ssw  secondary-structure weight */
  int32_t ss_mode;  /* par.ssm  (2 = Hit::SCORE_ALIGNMENT) */
} hhv_params;

/* one lane of the reference's ViterbiResult (src/hhviterbi.h:21-32) */
typedef struct {
  float score;
  int32_t i2;
  int32_t j2;
  int32_t index; /* template index inside the set */
} hhv_result;

/* the Hit fields ViterbiConsumerThread::align fills (src/hhviterbirunner.cpp:35-62) */
typedef struct {
  float score;          /* BacktraceScore.score: Viterbi score - score_ss + corr * Scorr */
  float viterbi_score;  /* raw ViterbiResult.score */
  float score_ss;       /* BacktraceScore.score_ss: sum of ScoreSS over the matched columns */
  int32_t index;        /* template index inside the set */
  int32_t i1, j1;       /* i_steps[nsteps], j_steps[nsteps] (alignment start) */
  int32_t i2, j2;       /* alignment end */
  int32_t nsteps;
  int32_t matched_cols;
} hhv_hit;

typedef struct hhv_ctx hhv_ctx;
typedef struct hhv_tset hhv_tset;

/* flags of hhv_align */
#define HHV_ALIGN_BACKTRACE 1u /* keep the backtrace bytes (needed by hhv_backtrace / hhv_hits / hhv_topk) */
#define HHV_ALIGN_CELLOFF 2u   /* honour the masks installed with hhv_set_celloff (implies BACKTRACE) */

int hhv_abi_version(void);
const char* hhv_last_error(void); /* thread local */

/* Host-side helpers (no device needed) that expose the packed layout of DESIGN.md section 2:
 * hhv_record_bytes() = size of one packed column record (112);
 * hhv_pack_profile(): index >= 0 -> header record + L column records with stream meta into
 * out[(L+1)*28]; index < 0 -> the L bare column records (the query layout) into out[L*28];
 * hhv_fast_log2_tables(): the lg2[1025]/diff[1025] tables of fast_log2 (src/util-inl.h:108-130). */
int32_t hhv_record_bytes(void);
int hhv_pack_profile(const float* p, const float* tr, int32_t L, int32_t index, float* out);
int hhv_fast_log2_tables(float* lg2, float* diff);

/* Number of usable HIP devices (0 and HHV_E_DEVICE when there is none). */
int hhv_device_count(int32_t* n);

/* Template-database sharding over the GPUs of one node: the unit of independence is the SIMD batch
 * loop of ViterbiRunner::alignment (src/hhviterbirunner.cpp:122, the OpenMP loop over batches), so whole templates are
 * distributed and no DP data crosses GPUs.  Templates are ordered by length descending (like :117-119), cut into bins
 * of 64 and the bins go to the currently least loaded shard (LPT on stream records = sum of L+1); equal lengths give
 * contiguous n/n_shards blocks.  shard_of[k] receives the shard of template k (0 .. n_shards-1).  Pure host function:
 * every rank of a multi-process job and the in-process multi-device runner (hhv::ShardedViterbiRunner) compute the same
 * plan from the same lengths. */
int hhv_shard_plan(int32_t n, const int32_t* L, int32_t n_shards, int32_t* shard_of);

/* The work queue of the DP kernel (replaces the static batch list of src/hhviterbirunner.cpp:117-122 - the
 * reference hands its sorted SIMD batches to OpenMP threads with schedule(dynamic), this is the device's form of it).  The
 * template stream (per template a header record + L columns) is cut into segments of whole templates in stream order, each
 * closed as soon as it holds >= HHV_SEGMENT_MIN_RECORDS records (a shorter remainder joins the last one); the systolic arrays
 * of the device draw them longest first (stable).  Which array aligns a template does not enter any result.  Pure host
 * function - what hhv_align uses internally, exposed so that the plan can be inspected and tested without a device.
 * seg: [2 * (n + 1)] (first record, end record) per segment in draw order, then the terminal entry (total, total + 1);
 * *n_seg = number of segments (<= n). */
#define HHV_SEGMENT_MIN_RECORDS 128
int hhv_segment_plan(int32_t n, const int32_t* L, int64_t* seg, int32_t* n_seg);

int hhv_create(hhv_ctx** out, const hhv_params* par);
/* Replace the context's fast_log2 tables (lg2[1025], diff[1025]; default: hhv_fast_log2_tables).  The reference keeps these
 * tables in function-local statics that the FIRST caller in the process initialises, and the initialiser is compiled per
 * translation unit (double log in one, logf in another): which of the two flavours a run sees depends on what ran first
 * (an .hhm query: HMM::AddTransitionPseudocounts; an alignment query: the alignment code).  A caller that lives in the same
 * process as the reference's code (the drop-in translation units) reads the tables the process actually has and hands them
 * over, so that per-column scores - and with them Hit.score - match to the last bit in every kind of run. */
int hhv_set_fast_log2_tables(hhv_ctx* ctx, const float* lg2, const float* diff);
/* Frees the context, its streams and the device blocks it keeps for reuse (the blocks of freed template sets are cached per
 * context, round 6).  The template, raw, prefilter and MAC sets of a context are freed BEFORE it (hhv_tset_free & co. return
 * their blocks to the context they came from). */
void hhv_destroy(hhv_ctx* ctx);
/* New search parameters for an existing context (par->device must be the context's device): what a process-wide
 * context that outlives one ViterbiRunner::alignment call needs (hh-suite_amd/dropin/hhviterbirunner_hip.cpp keeps the
 * raw template database of earlier searches resident in one).  The query, the resident sets and their results stay. */
int hhv_set_params(hhv_ctx* ctx, const hhv_params* par);
/* How the library launches the DP of a context - nothing a result depends on; for tests and measurements (the reference has
 * no counterpart: its batches are OpenMP iterations, src/hhviterbirunner.cpp:122).
 *   pair_mode      queries of two and more strips: -1 the library chooses (default), 0 one launch per strip, 1 two-wave
 *                  workgroups (pair / chain launches) wherever a pair kernel exists
 *   pair_swap      pair kernels: workgroups whose number has this bit set run the strips on swapped wave indices (default 0;
 *                  -1 none; at most 31)
 *   blocks_per_cu  > 0: at most this many resident workgroups per CU; 0 = what the kernel admits
 *   trace_mode     the backtrace walk (Viterbi::Backtrace, src/hhviterbi.cpp:83-160): -1 the library's choice (default; since
 *                  round 5 that is 1 at every set size - it measured ahead everywhere), 0 one lane per template,
 *                  1 one wavefront per template (a round trip per run of the path)
 * The defaults of a new context can be preset through the environment, read once in hhv_create: HHV_PAIR (0 / 1),
 * HHV_PAIR_SWAP, HHV_BLOCKS_PER_CU, HHV_TRACE_WAVE (0 / 1). */
int hhv_set_launch_policy(hhv_ctx* ctx, int32_t pair_mode, int32_t pair_swap, int32_t blocks_per_cu, int32_t trace_mode);
/* The context's device error word, read WITHOUT waiting for the stream (for callers that synchronise themselves): HHV_OK, or
 * HHV_E_DEVICE with the cause in hhv_last_error(); the word is cleared.  Reference analogue: the reference stops the process on an
 * illegal backtrace state (src/hhviterbi.cpp:139-144). */
int hhv_check_error(hhv_ctx* ctx);

/* query: p[(Lq+1)*20], tr[(Lq+1)*7] (copied before the call returns: the caller may reuse its arrays at once).
 * The call packs the rows into a pinned staging block; the next hhv_align / hhv_align_async moves the block to the device with
 * one kernel on the context's stream (no copy operation, no event: on the stream of a search loop each of those costs more than
 * the 58 KB do).  Device block and staging are kept between queries, and the call waits only for the PREVIOUS query's upload
 * kernel, so a loop that sets one query per search does not wait for the device.  Setting a second query before any alignment
 * replaces the first; hhv_hits behind an alignment works with the query of that alignment even if the next one - of the same
 * length: another length is refused with HHV_E_STATE - has been set.  HHV_QUERY_COPY=1 in the environment
 * of hhv_create: hipMemcpyAsync + event at once, as in the earlier rounds. */
int hhv_set_query(hhv_ctx* ctx, const float* p, const float* tr, int32_t Lq);

/* secondary structure (all optional; without them the engine runs the reference's no-SS kernels).
 * tables: S73[8][4][11], S33[4][11][4][11], S37[4][11][8] (NDSSP, NSSPRED, MAXCF of src/hhdecl.h:53-55);
 * query arrays [Lq+1] (call after hhv_set_query; NULL = absent);
 * mode: the ss_hmm_mode Viterbi::Align receives - 0 HMM::NO_SS_INFORMATION, 1 PRED_DSSP, 2 DSSP_PRED,
 * 4 PRED_PRED (src/hhhmm.h:58-61).  Like the reference, the SS term enters the DP only when
 * par.ss_mode == 2 (src/hhviterbi.cpp:175); hhv_hits always reports score_ss for the mode. */
int hhv_set_ss_tables(hhv_ctx* ctx, const float* S73, const float* S33, const float* S37);
int hhv_set_query_ss(hhv_ctx* ctx, const int8_t* ss_pred, const int8_t* ss_conf, const int8_t* ss_dssp);
int hhv_set_ss_mode(hhv_ctx* ctx, int32_t ss_hmm_mode);

/* n prepared template profiles -> resident packed set in HBM.  L[k] >= 1. */
int hhv_upload_templates(hhv_ctx* ctx, int32_t n, const int32_t* L, const float* const* p, const float* const* tr,
                         hhv_tset** out);
/* Same with secondary-structure records: ss_pred[k] / ss_conf[k] / ss_dssp[k] are [L[k]+1] arrays in the
 * reference's encoding (HMM::ss_pred 0..3, ss_conf 0..10, ss_dssp 0..7, src/hhhmm.h:151-154); the array of
 * pointers or any entry may be NULL (= no such record, zeros). */
int hhv_upload_templates_ss(hhv_ctx* ctx, int32_t n, const int32_t* L, const float* const* p, const float* const* tr,
                            const int8_t* const* ss_pred, const int8_t* const* ss_conf, const int8_t* const* ss_dssp,
                            hhv_tset** out);
/* Adopt an already packed DEVICE buffer (zero copy).  d_records holds the record stream described
 * in DESIGN.md section 2: for each template a header record followed by L[k] column records, one
 * terminal header, then >= HHV_STREAM_PAD records of slack; 28 floats per record.  The buffer must
 * outlive the set. */
#define HHV_STREAM_PAD 256
int hhv_adopt_device_stream(hhv_ctx* ctx, int32_t n, const int32_t* L, const void* d_records, hhv_tset** out);
/* ---- on-device template preparation ------------------------------------------------
 * Raw templates = the HMM fields HMM::Read leaves (src/hhhmm.cpp:202-694), before any pseudocounts:
 *   f[k]    [(L+2)*20]  match-state frequencies, rows 0..L+1 (rows 0 and L+1 unused)
 *   tr[k]   [(L+1)*7]   log2 transitions, enum order M2M,M2I,M2D,I2M,I2I,D2M,D2D
 *   neff[k] [(L+1)*3]   Neff_M, Neff_I, Neff_D per column;   neff_hmm[k] = Neff_HMM
 * hhv_prepare_templates runs PrepareTemplateHMM (src/hhfunc.cpp:165-202, HHM format: transition
 * pseudocounts, substitution-matrix amino-acid pseudocounts, background, null model) for every template on
 * the device and produces a resident set for hhv_align; with columnscore = 1 (the default) the result depends
 * on the query's average composition q_pav (HMM::pav after PrepareQueryHMM), so it is called once per query.
 * *out == NULL creates the set, otherwise the set created by an earlier call for the same raw set is refilled. */
typedef struct hhv_rawset hhv_rawset;
typedef struct {
  float gapd, gape, gapf, gapg, gaph, gapi, gapb; /* par.gap*        defaults 0.15 1 .6 .6 .6 .6 1 (src/hhdecl.cpp:74-80) */
  int32_t pcm;                                    /* par.pc_hhm_nocontext_mode  (0, 1, 2)            (src/hhdecl.cpp:64) */
  float pca, pcb, pcc;                            /* par.pc_hhm_nocontext_a/b/c 1.0 1.5 1.0 (pcc != 1: the host evaluates powf per column once) */
  int32_t columnscore;                            /* par.columnscore 0..3, default 1                 (src/hhdecl.cpp:98) */
  float pb[20];                                   /* background frequencies  (SetSubstitutionMatrix, src/hhmatrices.cpp:53-58) */
  float R[400];                                   /* R[a][b] = P(a|b)        (src/hhmatrices.cpp:66-69) */
} hhv_prep_params;
int hhv_upload_raw_templates(hhv_ctx* ctx, int32_t n, const int32_t* L, const float* const* f, const float* const* tr,
                             const float* const* neff, const float* neff_hmm, const int8_t* const* ss_pred,
                             const int8_t* const* ss_conf, const int8_t* const* ss_dssp, hhv_rawset** out);
void hhv_rawset_free(hhv_rawset* rs);
/* Raw template database file (the N1 idea for the N2 path): lengths, Neff_HMM and the raw column block exactly as it
 * sits in HBM; written once from the parsed .hhm files, opened per search without text parsing or repacking. */
int hhv_rawdb_write(const char* path, int32_t n, const int32_t* L, const float* const* f, const float* const* tr,
                    const float* const* neff, const float* neff_hmm, const int8_t* const* ss_pred,
                    const int8_t* const* ss_conf, const int8_t* const* ss_dssp);
int hhv_rawdb_open(hhv_ctx* ctx, const char* path, hhv_rawset** out);
int32_t hhv_rawset_size(const hhv_rawset* rs);
int hhv_rawset_lengths(const hhv_rawset* rs, int32_t* L);
/* Would hhv_prepare_templates / hhv_prepare_subset accept these parameters (HHV_OK), or are they outside what the device
 * preparation covers (HHV_E_LIMIT with text: pcm, columnscore, admixtures that could leave [0, 1])?  No device needed: a caller
 * that can prepare on the host instead (the drop-in) asks before it uploads raw templates - ONE definition of the limits. */
int hhv_prep_params_check(const hhv_prep_params* par);
int hhv_prepare_templates(hhv_ctx* ctx, hhv_rawset* rs, const hhv_prep_params* par, const float* q_pav, hhv_tset** out);
/* The same for a SUBSET of the resident raw set - the templates the prefilter let through: ids[n_ids] (any order,
 * repeats allowed) -> a NEW template set of n_ids templates in that order (template k of the set = raw template ids[k]),
 * to be released with hhv_tset_free.  Nothing is copied between host and device but the id list. */
int hhv_prepare_subset(hhv_ctx* ctx, hhv_rawset* rs, const hhv_prep_params* par, const float* q_pav, const int32_t* ids,
                       int32_t n_ids, hhv_tset** out);
/* average composition pav[n*20] of the prepared templates of the last hhv_prepare_templates (diagnostics/tests) */
int hhv_rawset_pav(hhv_ctx* ctx, hhv_rawset* rs, float* pav);
/* the prepared packed records of template k of a set ((L[k]+1)*28 floats: header + columns), device -> host */
int hhv_tset_records_of(hhv_ctx* ctx, hhv_tset* ts, int32_t k, float* out);
/* the whole record stream of a set in one copy: hhv_tset_records(ts) * 28 floats (template k starts at record
 * sum_{m<k} (L[m] + 1); the last record is the terminal header) */
int hhv_tset_download(hhv_ctx* ctx, hhv_tset* ts, float* out);

/* ---- HHblits prefilter kernels ---------------------------------------------------------
 * Prefilter::ungapped_sse_score / Prefilter::swStripedByte (src/hhprefilter.cpp:214-278, 70-212) of one query
 * column-state profile against a resident database of column-state (cs219) sequences.
 *   db:       n_db sequences concatenated, bytes 0..219 (219 = the ANY state), offsets[n_db+1]
 *   profile:  plain [220][Lq] bytes - the values Prefilter::stripe_query_profile computes (:386-425), before striping
 *   gapped=0: ungapped score with score_offset;  gapped=1: Smith-Waterman with gap_init (= gap open + extend),
 *             gap_extend and bias = score_offset, scores identical to the AVX2 build of the reference (32-byte stripes)
 *   subset:   nullable list of sequence ids (the survivors of the first filter); scores[n_subset or n_db] on the host */
typedef struct hhv_pfdb hhv_pfdb;
int hhv_prefilter_upload_db(hhv_ctx* ctx, int32_t n_db, const uint8_t* seqs, const int64_t* offsets, hhv_pfdb** out);
void hhv_prefilter_free_db(hhv_pfdb* db);
int hhv_prefilter_scores(hhv_ctx* ctx, hhv_pfdb* db, const uint8_t* profile, int32_t Lq, int32_t score_offset,
                         int32_t gapped, int32_t gap_init, int32_t gap_extend, const int32_t* subset, int32_t n_subset,
                         int32_t* scores);

/* ---- MAC realignment ------------------------------------------------------------------------
 * PosteriorDecoder::realign (src/hhposteriordecoder.cpp:86-119) for a batch of n hits of one query: forward, backward,
 * posterior, maximum-accuracy DP and MAC backtrace (src/hhforwardalgorithm.cpp, hhbackwardalgorithm.cpp,
 * hhmacalgorithm.cpp, hhbacktracemac.cpp), one wavefront per hit, bit-exact against the reference's doubles.
 *   q_p [(Lq+1)][20], q_tr_lin [(Lq+1)][7]: the prepared query with LINEAR transitions, i.e. after
 *       Log2LinTransitionProbs(1.0) and initializeQueryHMMTransitions (src/hhposteriordecoderrunner.cpp:146-155)
 *   t_p[k] [(Lt+1)][20], t_tr_lin[k] [(Lt+1)][7]: the prepared template of hit k, linear transitions, with the boundary
 *       assignments of initializeForAlignment (src/hhposteriordecoder.cpp:159-167)
 *   celloff[k]: (Lq+1)*(Lt+1) bytes, non-zero = cell excluded (the mask realign() builds: band around the Viterbi
 *       path, earlier alternative alignments, -excl regions); NULL = no cell excluded
 *   local / shift / mact: par.loc, par.shift, par.mact.  Secondary-structure scoring: hhv_mac_set_ss before the call.
 * hits[k] receives the alignment summary; paths and posteriors stay on the device in *out until fetched. */
typedef struct hhv_mac_hit {
  double Pforward;       /* Hit::Pforward (scaled total forward probability) */
  float sum_of_probs;    /* Hit::sum_of_probs */
  int32_t i1, j1, i2, j2, nsteps, matched_cols;
  int32_t reserved;
} hhv_mac_hit;
typedef struct hhv_macset hhv_macset;
int hhv_mac_realign(hhv_ctx* ctx, const float* q_p, const float* q_tr_lin, int32_t Lq, int32_t n, const int32_t* Lt,
                    const float* const* t_p, const float* const* t_tr_lin, const uint8_t* const* celloff, int32_t local,
                    float shift, float mact, hhv_macset** out, hhv_mac_hit* hits);
/* The same with the masks built ON THE DEVICE from what realign() derives them from (src/hhposteriordecoder.cpp:92-109):
 * the Viterbi alignment of the hit (end points, path entries 1..nsteps of Hit::i / Hit::j), the cells of the MAC
 * alignments found earlier for the same template (Hit::alt_i / alt_j, concatenated) and the -excl / -template_excl
 * ranges ((lo, hi) pairs, already made absolute).  No (Lq+1)*(Lt+1) byte mask crosses the bus. */
typedef struct hhv_mac_input {
  int32_t i1, j1, i2, j2;
  int32_t nsteps;
  int32_t n_excluded;
  const int32_t* i;           /* [nsteps+1], entry 0 unused */
  const int32_t* j;
  const int32_t* excluded_i;  /* [n_excluded] */
  const int32_t* excluded_j;
} hhv_mac_input;
int hhv_mac_realign_hits(hhv_ctx* ctx, const float* q_p, const float* q_tr_lin, int32_t Lq, int32_t n, const int32_t* Lt,
                         const float* const* t_p, const float* const* t_tr_lin, const hhv_mac_input* in, int32_t n_qranges,
                         const int32_t* qranges, int32_t n_tranges, const int32_t* tranges, int32_t local, float shift,
                         float mact, hhv_macset** out, hhv_mac_hit* hits);
/* The same with the template PROFILES read from a resident template set (the one the Viterbi stage just searched: hit k is
 * template template_of[k] of ts) - only the linear transitions of the realigned templates (7 floats per column; powf is
 * the host's, see hhv::LinearTransitions) are handed over.  An input with i == NULL and j == NULL takes its Viterbi
 * alignment (end points and path) from the set's own trace results (the last hhv_hits on ts), so that nothing of the
 * Viterbi stage has to be fetched to realign its best hits. */
int hhv_mac_realign_tset(hhv_ctx* ctx, const float* q_p, const float* q_tr_lin, int32_t Lq, hhv_tset* ts, int32_t n,
                         const int32_t* template_of, const float* const* t_tr_lin, const hhv_mac_input* in, int32_t n_qranges,
                         const int32_t* qranges, int32_t n_tranges, const int32_t* tranges, int32_t local, float shift,
                         float mact, hhv_macset** out, hhv_mac_hit* hits);
/* Secondary-structure scoring inside the NEXT hhv_mac_realign* call of this context: PosteriorDecoder multiplies the match
 * probability of every cell by fpow2(Viterbi::ScoreSS(q, t, i, j, ssw, hit.ssm2, ...)) (src/hhforwardalgorithm.cpp:77,100,
 * src/hhbackwardalgorithm.cpp:82).
 *   tables  [2][352] floats, the factors themselves: [0] = fpow2(ssw * S37[q_pred][q_conf][t_dssp]) as [44][8] (hit mode 1,
 *           HMM::PRED_DSSP), [1] = fpow2(ssw * S73[q_dssp][t_pred][t_conf]) as [8][44] (mode 2, HMM::DSSP_PRED)
 *   q_idx   [2][Lq+2] row index of every query column for the two modes (ss_pred*11 + ss_conf; ss_dssp)
 *   mode[k] hit.ssm2 of hit k: 0 = none, 1, 2 (ssm2 = 3 scores nothing in the reference: pass 0)
 *   t_idx[k] [Lt[k]+2] column index of the template for the hit's mode (ss_dssp; ss_pred*11 + ss_conf); entry Lt+1 is the
 *           element PAST the template, which the reference reads for the cells of column 1 (the stale loop variable of
 *           src/hhforwardalgorithm.cpp:77).  NULL for mode 0.
 * The setting is consumed by one call.  Templates of any length. */
int hhv_mac_set_ss(hhv_ctx* ctx, const float* tables, const uint8_t* q_idx, int32_t Lq, int32_t n, const int32_t* mode,
                   const uint8_t* const* t_idx, const int32_t* Lt);
/* the mask of hit k as the kernels saw it, (Lq+1)*(Lt+1) bytes */
int hhv_mac_celloff(hhv_macset* ms, int32_t k, uint8_t* mask);
/* path of hit k: entries 1..nsteps (Hit::i, ::j, ::states, ::S, ::P_posterior); cap >= nsteps + 1 */
int hhv_mac_path(hhv_macset* ms, int32_t k, int32_t cap, int32_t* i_steps, int32_t* j_steps, int8_t* states, float* S,
                 float* P_posterior, int32_t* nsteps);
/* dense posterior matrix of hit k, (Lq+1)*(Lt+1) floats (row 0 / column 0 unused) */
int hhv_mac_posterior(hhv_macset* ms, int32_t k, float* posterior);
/* The sparse lists the reference attaches to a realigned hit for its -o_matrices output (PosteriorDecoder::
 * writeProfilesToHits, src/hhbacktracemac.cpp:14-110; entries pushed by src/hhforwardalgorithm.cpp:184-219 and
 * src/hhbackwardalgorithm.cpp:112-122): which = 0 forward (Hit::forward_matrix), 1 backward (Hit::backward_matrix),
 * 2 posterior (Hit::posterior_matrix: posterior >= 0.01, finite, cell on - the mask of the DP plus the cells within two rows /
 * columns of a step of the MAC path, which backtraceMAC switches off before the list is built).  Entries (i, j, value) in the reference's order
 * (i ascending, then j).  Returns the number of entries of the list (also when cap is smaller: call again with room), or
 * a negative error.  Lists 0 and 1 exist only for sets computed after hhv_mac_set_lists(ctx, 1) - the setting holds until it is
 * changed and costs two more matrices per hit on the device; Hit::forward_profile[i] / backward_profile[i] are the sums of a
 * list's values of row i in list order. */
int hhv_mac_set_lists(hhv_ctx* ctx, int32_t on);
int64_t hhv_mac_list(hhv_macset* ms, int32_t k, int32_t which, int64_t cap, int32_t* i, int32_t* j, float* value);
void hhv_macset_free(hhv_macset* ms);

/* A new template set made of templates ids[0..n) of a resident one (any order, repeats allowed), copied on the device:
 * the surviving templates of an alternative-alignment round (src/hhviterbirunner.cpp:260-268) or any selection of a
 * resident database, without touching host memory.  Template k of the new set is template ids[k] of ts. */
int hhv_tset_gather(hhv_ctx* ctx, hhv_tset* ts, const int32_t* ids, int32_t n, hhv_tset** out);

/* The whole first stage of Prefilter::prefilter_db on the device (src/hhprefilter.cpp:461-505): gapless scores of ALL
 * sequences, length correction score - (int)(bit_factor * (log_qlen + flog2(len))) (log_qlen = flog2(Lq), util-inl.h:83),
 * descending sort by (score, id), keep the min_hits best plus everything above smax_thresh.  ids[0..*n_out) = the
 * surviving sequence ids, best first - the same set and order hhv::Prefilter::SelectFirst derives from the scores on
 * the host; only the ids come back (4 bytes per survivor instead of 4 bytes per database sequence). */
int hhv_prefilter_first(hhv_ctx* ctx, hhv_pfdb* db, const uint8_t* profile, int32_t Lq, int32_t score_offset, float log_qlen,
                        int32_t bit_factor, int32_t smax_thresh, int32_t min_hits, int32_t* ids, int32_t cap, int32_t* n_out);

/* Binary packed template database: the record stream plus its length table in one file, so that
 * a search mmaps/reads it straight into HBM instead of parsing and re-packing HMM text per query.
 * File = 64-byte header {magic "HHVPDB01", int32 n, int32 record_dwords (28), int64 n_records, zero pad},
 * int32 L[n], then n_records * 28 floats (header/column records incl. the terminal header).
 * hhv_db_write packs host profiles to `path`; hhv_db_open loads a file into a resident set. */
int hhv_db_write(const char* path, int32_t n, const int32_t* L, const float* const* p, const float* const* tr,
                 const int8_t* const* ss_pred, const int8_t* const* ss_conf, const int8_t* const* ss_dssp);
int hhv_db_open(hhv_ctx* ctx, const char* path, hhv_tset** out);
void hhv_tset_free(hhv_tset* ts);
int32_t hhv_tset_size(const hhv_tset* ts);
int64_t hhv_tset_cells(const hhv_tset* ts, int32_t Lq); /* sum over templates of Lq*L[k] */
/* number of stream records (sum(L+1)+1) and size of one packed record in bytes */
int64_t hhv_tset_records(const hhv_tset* ts);

/* Viterbi::Align for every template of the set.  out (host, n entries, nullable) receives
 * score/i2/j2 in template order.  Synchronous. */
int hhv_align(hhv_ctx* ctx, hhv_tset* ts, uint32_t flags, hhv_result* out);
/* Same, asynchronous on the context's stream, results stay on the device: d_out is a DEVICE
 * pointer to n hhv_result (nullable = internal buffer only).  hhv_sync waits. */
int hhv_align_async(hhv_ctx* ctx, hhv_tset* ts, uint32_t flags, void* d_out);
int hhv_sync(hhv_ctx* ctx);
/* the HIP stream (hipStream_t) the context launches on, for callers that time with HIP events */
void* hhv_stream(hhv_ctx* ctx);
/* duration in milliseconds of the last hhv_align* DP kernel launch, from HIP events recorded
 * around the launch on the context's stream (valid after hhv_sync) */
int hhv_last_kernel_ms(hhv_ctx* ctx, float* ms);

/* cell-off mask of template k for the next hhv_align with HHV_ALIGN_CELLOFF: mask[(Lq+1)*(L[k]+1)],
 * non-zero = excluded.  NULL clears. */
int hhv_set_celloff(hhv_ctx* ctx, hhv_tset* ts, int32_t k, const uint8_t* mask);
/* The masks of a whole alternative-alignment round built on the device (exclude_alignments -> Viterbi::ExcludeAlignment,
 * src/hhviterbirunner.cpp:152-164,273-289; src/hhviterbi.cpp:61-77): every mask of ts is cleared, then each of the n_paths
 * earlier alignments (path p belongs to template template_of[p]; steps path_off[p] .. path_off[p+1]-1 of i_steps/j_steps,
 * i.e. entries 1..nsteps of Hit::i / Hit::j - the last one is skipped like the reference does) switches off its +-40 cross,
 * and the -excl / -template_excl (lo, hi) ranges are applied.  Replaces one hhv_set_celloff (a (Lq+1)*(Lt+1) byte mask
 * built and copied by the host) per surviving template.  The arrays are copied before the call returns (the caller may reuse
 * them at once); the copies and the kernels are only enqueued on the context's stream - the call does not wait for the device. */
int hhv_set_celloff_paths(hhv_ctx* ctx, hhv_tset* ts, int32_t n_paths, const int32_t* template_of, const int64_t* path_off,
                          const int32_t* i_steps, const int32_t* j_steps, int32_t n_qranges, const int32_t* qranges,
                          int32_t n_tranges, const int32_t* tranges);
/* Global mode only (par.loc = 0): reproduce the reference's SIMD-batch behaviour.  Viterbi::Align maximises the global score
 * over the last row and over the last column OF THE BATCH, i.e. of the longest of the <= VECSIZE_FLOAT templates aligned
 * together (src/hhviterbialgorithm.cpp:462-486); for a shorter template of the batch that column is padding, so only its last
 * ROW counts.  not_longest[k] != 0 marks template k as such a template (its own last column is then left out of the
 * maximisation); NULL clears all marks = every template as if aligned alone (HMMSimd::MapOneHMM), the default.  The marks
 * stay with the set until set again; sets made by hhv_tset_gather inherit those of their source. */
int hhv_set_global_batch(hhv_ctx* ctx, hhv_tset* ts, const uint8_t* not_longest);
/* raw backtrace byte matrix of template k in the reference layout: out[(Lq+1)*(L[k]+1)] */
int hhv_backtrace_matrix(hhv_ctx* ctx, hhv_tset* ts, int32_t k, uint8_t* out);

/* Viterbi::Backtrace + Viterbi::ScoreForBacktrace for all templates, on the device (needs a
 * preceding hhv_align with HHV_ALIGN_BACKTRACE).  hits (host, n entries, nullable; NULL = the kernels are only
 * enqueued on the context's stream, the records stay on the device for hhv_topk / hhv_hit_path). */
int hhv_hits(hhv_ctx* ctx, hhv_tset* ts, hhv_hit* hits);
/* Viterbi::Backtrace(matrix, elem, i2, j2) -> BacktraceResult (src/hhviterbi.cpp:83-160, src/hhviterbi.h:34-40) of template k:
 * i_steps / j_steps / states of cap entries (nullable), 1-based, step 1 = alignment END, states[nsteps] = MM like the
 * reference; *nsteps = BacktraceResult.count, *matched_cols (nullable) = BacktraceResult.matched_cols.  Needs a preceding
 * hhv_align with HHV_ALIGN_BACKTRACE.  The device walks the paths of all templates of the set in one launch (the first call
 * after an alignment starts it, like hhv_hits); per-step scores and Hit fields: hhv_hits / hhv_hit_path. */
int hhv_backtrace(hhv_ctx* ctx, hhv_tset* ts, int32_t k, int32_t cap, int32_t* i_steps, int32_t* j_steps, int8_t* states,
                  int32_t* nsteps, int32_t* matched_cols);
/* path of template k: arrays of cap entries, 1-based like BacktraceResult (index 0 unused,
 * step 1 = alignment end); S = per-step column scores (BacktraceScore.S).  Needs hhv_hits. */
int hhv_hit_path(hhv_ctx* ctx, hhv_tset* ts, int32_t k, int32_t cap, int32_t* i_steps, int32_t* j_steps,
                 int8_t* states, float* S, int32_t* nsteps);
/* All paths of the last hhv_hits at once: pointers into the host mirror of the path pool, owned by the set and valid
 * until the set is aligned again or freed.  Path of template k = entries path_off[k] + 1 .. path_off[k] + nsteps of the four
 * arrays (entry path_off[k] is the unused index 0 of BacktraceResult).  For callers that build thousands of Hit objects:
 * one call instead of one hhv_hit_path per hit. */
int hhv_hit_path_pool(hhv_ctx* ctx, hhv_tset* ts, const int64_t** path_off, const int32_t** i_steps, const int32_t** j_steps,
                      const int8_t** states, const float** S);
/* The paths of the last hhv_hits WITHOUT the pool's unused capacity: one compact record stream, entries off[k] + 0 .. off[k] +
 * hits[k].nsteps of hit k (entry 0 is the unused index 0 of BacktraceResult, all zero), i and j as 16-bit values.  hits = what
 * hhv_hits returned for this set (its step counts size the records).  The arrays live in a pinned buffer of the CONTEXT and stay
 * valid until the next hhv_hit_paths_packed call of any set of the context.  For callers that turn every path into a Hit
 * (ViterbiRunner::alignment, src/hhviterbirunner.cpp:35-67): 9 bytes per path step cross PCIe instead of 13 per pool entry. */
int hhv_hit_paths_packed(hhv_ctx* ctx, hhv_tset* ts, const hhv_hit* hits, const int64_t** off, const uint16_t** i_steps,
                         const uint16_t** j_steps, const int8_t** states, const float** S);
/* K best hits by hit score (descending, ties by smaller index), selected on the device.
 * flags: 0 = rank by Hit.score (needs hhv_hits); HHV_TOPK_RAW = rank by the raw Viterbi score of the
 * last hhv_align (score-only searches: the records carry viterbi_score, i2, j2, index; path fields 0).
 * out: host, k entries (nullable); d_out: DEVICE pointer to k hhv_hit records (nullable) - the buffer a
 * multi-GPU caller hands to its all-gather; entries beyond *n_out are filled with 0xFF bytes.
 * With out == NULL the call only enqueues work on the context's stream (hhv_stream) and does not wait for it: a caller
 * that consumes d_out on another stream orders the two with an event (bench.py: hhv_topk -> all_gather -> hhv_merge_hits). */
#define HHV_TOPK_RAW 1u
/* HHV_TOPK_PVALUE: rank by the reference's own sort key instead of Hit.score - score_aass, which HitList::CalculatePvalues
 * (src/hhhitlist.cpp:499-531) derives from score and score_ss through the extreme-value distribution of the pair's lengths and
 * diversities (lamda_NN / mu_NN, src/hhhitlist-inl.h:13-66; Hit::CalcEvalScoreProbab, src/hhhit.h:134-141; Hit::operator<,
 * src/hhhit.h:116-126).  For sharded searches: a shard's K-cut by Hit.score can drop a hit the reference ranks inside the top K.
 * Needs hhv_hits and hhv_tset_set_neff; the records come out in that order, their fields unchanged (the host recomputes the
 * reference's numbers for them - the device's exp / log are within an ulp of libm's, not bit for bit). */
#define HHV_TOPK_PVALUE 2u
int hhv_topk(hhv_ctx* ctx, hhv_tset* ts, int32_t k, uint32_t flags, hhv_hit* out, void* d_out, int32_t* n_out);
/* Sharded databases (one hhv_ctx / hhv_tset per GPU, hhv_shard_plan): ids[k] >= 0 = the GLOBAL template id of entry k
 * of this shard (host array, ts->n entries, copied).  From then on hhv_topk reports ids[index] in hhv_hit.index, so that
 * its output can go straight into the exchange; NULL switches back to the index inside the set. */
int hhv_tset_set_global_ids(hhv_ctx* ctx, hhv_tset* ts, const int32_t* ids);
/* The diversities HHV_TOPK_PVALUE needs: q_neff = q->Neff_HMM, t_neff[k] = Neff_HMM of template k (HMM::Neff_HMM, src/hhhmm.h; the
 * template lengths are the set's).  Host array of ts->n entries, copied. */
int hhv_tset_set_neff(hhv_ctx* ctx, hhv_tset* ts, float q_neff, const float* t_neff);
/* The merge half of the sharded top-K: d_in = DEVICE pointer to m hhv_hit records, the concatenation of every shard's
 * hhv_topk output after the all-gather (records with index < 0 are padding).  Returns the k best by score (descending,
 * ties by the smaller global id - the order the reference's caller gives the hit list, src/hhhit.h:116-126, after
 * ViterbiRunner::alignment appended the batches one after the other, src/hhviterbirunner.cpp:173); identical on every
 * rank.  out: host, k entries (nullable); d_out: DEVICE, k entries (nullable); entries beyond *n_out are 0xFF bytes.
 * With out == NULL and n_out == NULL the merge is only enqueued on the context's stream (no host wait). */
int hhv_merge_hits(hhv_ctx* ctx, const void* d_in, int32_t m, int32_t k, hhv_hit* out, void* d_out, int32_t* n_out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HHVITERBI_HIP_H */
