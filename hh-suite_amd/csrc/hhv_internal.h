// hhv_internal.h -- structures shared between the HIP kernels and the C-ABI host layer.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace hhv {

constexpr int LANES = 64;          // wavefront width on gfx950
constexpr int MAX_R = 5;           // query rows per lane (<= 256 VGPRs, 2 waves/SIMD); one pass covers 64*R rows
constexpr int CHUNK_RECS = 32;     // records per LDS refill (3.5 x 1 KiB global_load_lds_dwordx4)
constexpr int RING_CHUNKS = 4;     // the 64-record live window spans <= 3 chunks, the fourth is in flight
constexpr int RING_RECS = CHUNK_RECS * RING_CHUNKS;
constexpr int STREAM_PAD_RECS = 256;  // slack after the terminal header (chunk over-read)
constexpr int BT_ENTRY_BYTES = 8;  // one backtrace entry = the R (<= 8) bytes of one lane at one column

// Strip plan of a query.
// Long queries: the Lq rows are cut into P passes of 64 * R_p rows.  With U = ceil(Lq / 64) row units, P = ceil(U / MAX_R)
// is the smallest number of passes and the units are spread as evenly as possible: the first P_hi passes take R_hi rows
// per lane, the others R_hi - 1 (Lq = 431: 4 + 3 rows per lane = 448 rows instead of 2 x 4 = 512; Lq = 2000: 4 x 5 + 3 x 4).
// Every pass is its own launch (R is a template parameter of the kernel).
// Short queries: a 64-lane array would leave most rows of the strip empty (Lq = 80: R = 2, 128 rows) and pay the per-step
// overhead for few cells, so the wavefront is split into 64 / W independent systolic arrays of W = 32 (Lq <= 160) or
// 16 (Lq <= 80) lanes, each walking its own range of the template stream with R = ceil(Lq / W) rows per lane.
struct StripPlan {
  int32_t P = 0, P_hi = 0, R_hi = 0;
  int32_t W = LANES;  // lanes per systolic array (64, 32 or 16; < 64 only with P == 1)
  __host__ __device__ int R(int p) const { return p < P_hi ? R_hi : R_hi - 1; }
  __host__ __device__ int base(int p) const {  // rows before pass p
    return p <= P_hi ? p * W * R_hi : P_hi * W * R_hi + (p - P_hi) * W * (R_hi - 1);
  }
  __host__ __device__ int rows() const { return base(P); }
  // 1-based query row i -> backtrace plane (pass), lane of the array, row of the lane, rows per lane of that pass
  __host__ __device__ void locate(int i, int& pass, int& lane, int& r, int& Rp) const {
    int k = i - 1;
    const int hi_rows = P_hi * W * R_hi;
    if (k < hi_rows) {
      Rp = R_hi;
      pass = k / (W * R_hi);
      k -= pass * W * R_hi;
    } else {
      Rp = R_hi - 1;
      k -= hi_rows;
      const int q = k / (W * Rp);
      pass = P_hi + q;
      k -= q * W * Rp;
    }
    lane = k / Rp;
    r = k - lane * Rp;
  }
  __host__ __device__ bool operator==(const StripPlan& o) const {
    return P == o.P && P_hi == o.P_hi && R_hi == o.R_hi && W == o.W;
  }
  // max_array_lanes: 64 disables the short-query arrays (HHV_ARRAY_LANES, for measurements)
  static StripPlan make(int Lq, int max_split = 16) {
    StripPlan s;
    s.W = LANES;
    if (Lq <= 16 * MAX_R && max_split <= 16) s.W = 16;
    else if (Lq <= 32 * MAX_R && max_split <= 32) s.W = 32;
    const int U = (Lq + s.W - 1) / s.W;
    s.P = (U + MAX_R - 1) / MAX_R;
    s.R_hi = (U + s.P - 1) / s.P;
    s.P_hi = U - s.P * (s.R_hi - 1);
    return s;
  }
};

// Backtrace / cell-off buffer: one plane per pass, one 8-byte entry per (stream record, lane).
// The entry of (record rec, lane g) sits at row rec + g, column g: at step s lane g works on record s - g, so the W
// entries a systolic array stores in one step share the row (first record + s) - ONE contiguous W * 8 byte store per
// array and step, whatever the wave partition is (a record-major layout would scatter the 8-byte pieces of a step over
// W different rows: 4 x write amplification at the HBM, measured with WRITE_SIZE in profiles/r1bt_summary.txt).
__host__ __device__ inline size_t bt_entry(int64_t rec, int g, int W) { return (size_t)(rec + g) * (size_t)W + (size_t)g; }
__host__ __device__ inline size_t bt_plane_entries(int64_t n_records, int W) { return (size_t)(n_records + W) * (size_t)W; }

struct DevResult {  // matches hhv_result
  float score;
  int32_t i2, j2;
  int32_t index;
};

struct DevHit {  // matches hhv_hit
  float score;
  float viterbi_score;
  float score_ss;
  int32_t index;
  int32_t i1, j1, i2, j2;
  int32_t nsteps;
  int32_t matched_cols;
};

struct StreamArgs {
  const float* records;      // [n_records + pad][28]
  const int64_t* wave_rec;   // [n_waves * (64 / W) + 1] first record of each array's template range
  const float* qpack;        // [64*R][28]
  DevResult* results;        // [n_templates]
  uint64_t* bt;              // backtrace / cell-off entries, bt_entry() layout (BT / CELLOFF variants)
  float negq, negt, shift;  // 0 - par.egq, 0 - par.egt (viterbi_lane.h Params), par.shift
  int32_t Lq;
  // multi-pass strips (StripPlan): pass p handles query rows row_base+1 .. row_base+64*R_p
  int32_t row_base;          // StripPlan::base(p)
  int32_t bt_plane;          // p: the plane of the backtrace buffer this pass reads masks from / writes to
  int32_t pass_first;        // lane 0 takes the DP boundary row 0 (else: the carry of the previous pass)
  int32_t pair_swap;         // pair kernels: workgroups whose number has bit pair_swap set run the strips on swapped wave indices (-1: none)
  int32_t pass_last;         // the lane owning row Lq emits the result (else: lane 63 writes the carry)
  float4* carry;             // [n_records] bottom-row state {MM,GD,IM,DG} of the previous / for the next pass ...
  float* carry_mi;           // ... and MI
  int64_t bt_pass_stride;    // entries per plane = bt_plane_entries(n_records, W)
  // secondary-structure term (SS variants): ss(i,j) = ss_table[ss_q_off[i-1] + ((meta_j >> ss_t_shift) & ss_t_mask)]
  const float* ss_table;     // ssw * S33 / S73 / S37, premultiplied on the host (same fp32 product as the reference)
  const int32_t* ss_q_off;   // [P*64*R] table row offset of query row i at index i-1
  int32_t ss_t_shift, ss_t_mask;
  int32_t ss_tab_n;          // floats of the table (1936 / 352): what hhv_ss_kernel copies into LDS
  // work queue of the 64-lane variants (hhv_stream_kernel.h DQ); unused by the short-query arrays (fixed ranges, wave_rec)
  const int64_t* seg_first;  // [n_seg + 1][2] records [first, end) of segment k in the order they are drawn: whole templates,
                             // >= 128 records, longest first; entry n_seg = the terminal header
  int32_t n_seg;
  uint32_t* queue;           // ticket counter, = the number of waves at launch (wave w starts with segment w)
  uint32_t* err;             // the context's device error word (host-mapped): DEV_ERR_* bits, set by system-scope atomics
};

// Device-side failures (the reference stops loudly on an illegal state, src/hhviterbi.cpp:139-144): a kernel that meets one
// sets a bit in the context's error word; every call that waits for the stream reports it as HHV_E_DEVICE (hhv_api.cpp).
constexpr uint32_t DEV_ERR_PAIR_TIMEOUT = 1u;   // a wave of a two-wave workgroup waited in vain for its partner (hhv_stream_kernel.h pair_wait)
constexpr uint32_t DEV_ERR_TRACE_STATE = 2u;    // hhv_trace_kernel met a state that is not one of STOP, MM, GD, IM, DG, MI
constexpr uint32_t DEV_ERR_MAC_TIMEOUT = 4u;    // a wavefront of a MAC forward / backward workgroup waited in vain for another one's progress counter (hhv_mac.hip df_wait)

struct TraceArgs {
  const float* records;
  const int64_t* rec_off;    // [n+1] header record of template k
  const int32_t* L;          // [n]
  const float* qp;           // query profile AoS [(Lq+1)][20]
  const DevResult* results;
  const uint64_t* bt;
  const float* lg2;          // fast_log2 tables (util-inl.h:108-130): lg2[1025], diff[1025]
  const float* diff;
  DevHit* hits;
  int32_t* i_steps;          // [path_off[k] .. ) per template, capacity Lq + L[k] + 2
  int32_t* j_steps;
  int8_t* states;
  float* S;
  float* Sss;                // per-step secondary-structure scores (same pools as S), null without secondary-structure information
  const int64_t* path_off;   // [n+1]
  float corr;
  int32_t ss_mode;
  int32_t Lq, n;
  StripPlan plan;            // which plane / lane / byte of the backtrace buffer holds query row i
  int64_t bt_pass_stride;    // backtrace entries per pass
  int32_t bt_mm;             // encoding of the MM predecessor in the entries (viterbi_lane.h bt_mm_mode of the launch that wrote them)
  const float* ss_table;     // null: no secondary-structure information (score_ss = 0)
  const int32_t* ss_q_off;
  int32_t ss_t_shift, ss_t_mask;
  uint32_t* err;             // the context's device error word (StreamArgs::err)
  int32_t trace_mode;        // -1 the launcher chooses, 0 one lane per template, 1 one wavefront per template (hhv_kernels.hip launch_trace)
};

// raw (unprepared) template column, 32 dwords: the fields of the reference's HMM after HMM::Read
constexpr int RAW_DW = 32;
constexpr int RAW_F = 0;      // [0..19]  f[i][a]
constexpr int RAW_TR = 20;    // [20..26] tr[i][7], enum order M2M,M2I,M2D,I2M,I2I,D2M,D2D
constexpr int RAW_NEFF = 27;  // [27..29] Neff_M[i], Neff_I[i], Neff_D[i]
constexpr int RAW_J = 30;     // int: column index i | (ss meta bits 16..24, see viterbi_lane.h) in RAW_SS
constexpr int RAW_L = 31;     // int: L of the template
constexpr int RAW_SS = 30;    // the secondary-structure bits share the dword with the column index

struct PrepArgs {
  const float* raw;          // [n_cols][32]
  int64_t n_cols;            // sum(L+1)
  const int64_t* rec_off;    // [n slots + 1] output records
  const int32_t* L;          // [n slots]
  const float* neff_hmm;     // [n raw]
  const float* pb;           // [20]
  const float* R;            // [20][20]
  const float* q_pav;        // [20]
  const float* lg2;
  const float* diff;
  float* p_tmp;              // [n_cols][20]
  float* tr_tmp;             // [n_cols][8]
  float* records;            // output stream
  float* pav_out;            // [n][20] or null
  float gapd, gape, gapf, gapg, gaph, gapi, gapb;
  int32_t pcm;
  float pca, pcb, pcc;       // (pcm 3: pca holds the reference's recomputed pca = 0.793 + 0.048 (pcb - 10), hhv_api_prep.cpp)
  const float* tau;          // pcm 2 with pcc != 1: tau of every raw column (the host evaluated libm's powf, hhv_api_prep.cpp); else null
  int32_t columnscore;
  const int32_t* ids;        // output slots of this launch (one workgroup each)
  const int32_t* src;        // [n slots] raw template of slot k, or null = identity (the whole raw set)
  const int64_t* raw_off;    // [n raw + 1] first column of every raw template in the raw block
  int32_t lds_cols;          // fused kernel: columns the LDS buffers hold (max L of the class + 1)
  int32_t step3_all;         // fused kernel: step 3 on every wavefront (else on those that had no step 2)
};

size_t prepare_fused_lds(int max_L);
int launch_prepare(const PrepArgs& a, const int32_t* const ids[3], const int32_t n_ids[3], const int32_t max_L[3], void* stream);
int launch_gather_neff(const float* raw, int64_t n_cols, float* out, void* stream);  // out[i] = Neff_M of raw column i

struct PrefilterArgs {
  const unsigned char* profile;  // plain [220][Lq]
  const unsigned char* striped;  // [220][W][32] like the reference's qc (generic kernel without LDS profile), or null
  const unsigned char* seqs;     // concatenated column-state sequences (4-byte aligned base, padded by 16 bytes)
  const int64_t* offsets;        // [n_db + 1]
  const int32_t* subset;         // [n_jobs] sequence ids, or null = all
  const int32_t* order;          // [n_jobs] job -> slot, longest sequence first, or null
  int32_t* scores;               // [n_jobs], indexed by slot
  int64_t n_jobs;
  int32_t Lq, W, offset, gap_init, gap_extend;
  // gapless kernel on a query cut into slabs: first query position of this launch, S of the previous slab's last row
  // (one byte per db residue) and where to leave this slab's last row; null / 0 for a query that fits one slab
  int32_t q_base;
  const unsigned char* carry_in;
  unsigned char* carry_out;
  unsigned char* state_scratch;  // generic kernel: H/E columns of queries beyond LDS, [block][8][3][W][32]; null = LDS
};
// fast kernels: W = cells per lane (ungapped: ceil(slab/64) <= 8, Smith-Waterman: ceil(Lq/32) <= 20), profile as int8 in LDS
size_t prefilter_fast_lds(bool gapped, int W);
int launch_prefilter_fast(const PrefilterArgs& a, bool gapped, int W, int n_blocks, void* stream);
// generic kernel: a.W = ceil(Lq/32); prof_lds = striped profile built in LDS, else read from a.striped
// cell-off masks of the alternative-alignment rounds from the earlier alignments' paths (hhv_topk.hip)
int celloff_from_paths(uint64_t* bt, const int64_t* rec_off, const int32_t* L, int64_t pass_stride, int Lq, StripPlan plan,
                       int n_templates, int n_paths, const int32_t* template_of, const int64_t* path_off, const int32_t* pi,
                       const int32_t* pj, const int32_t* ranges, int n_q, int n_t, int max_Lt, hipStream_t stream);
// device-side subset of a resident template set (hhv_topk.hip)
int set_header_flags(float* records, const int64_t* rec_off, const unsigned char* flags, int n, hipStream_t stream);
int tset_gather(const float* src, const int64_t* src_off, const int32_t* ids, const int64_t* dst_off, const int32_t* L, int n,
                float* dst, hipStream_t stream);
// first selection step of the prefilter on the device (hhv_topk.hip)
size_t topk_temp_bytes(int n);
int pf_select_sort(const int32_t* d_scores, const int64_t* d_offsets, int n, float log_qlen, int bit_factor, int smax_thresh,
                   uint64_t* keys, uint64_t* sorted, void* temp, size_t temp_bytes, unsigned int* above, hipStream_t stream);
int pf_select_ids(const uint64_t* sorted, int m, int32_t* d_ids, hipStream_t stream);
int launch_prefilter_generic(const PrefilterArgs& a, bool gapped, bool prof_lds, int n_blocks, size_t lds_bytes, void* stream);

// MAC realignment (hhv_mac.hip): one wavefront per hit
struct DevMacHit {
  double Pforward;
  float sum_of_probs;
  int32_t i1, j1, i2, j2, nsteps, matched_cols, pad;
};
struct MacArgs {
  int32_t n, Lq;
  const float* q_p;          // [(Lq+1)][20]
  const float* q_tr;         // [(Lq+1)][7] linear
  const float* t_p;          // template columns: staged [col][20], or the record stream of a resident set ([col][28])
  int32_t t_p_stride;        // 20 or 28
  const int64_t* p_off;      // [n] first column of hit k in t_p when it differs from col_off (resident set), else null
  const float* t_tr;         // [col][7] linear
  const int64_t* col_off;    // [n] first column (index 0) of hit k
  const int32_t* Lt;         // [n]
  const int64_t* mat_off;    // [n] offset of the (Lq+1)*(Lt+1) matrices of hit k
  const unsigned char* celloff;
  float* mat;                // F_MM, then posterior
  unsigned char* bmm;        // MAC backtrace codes
  double* scale;             // [n][Lq+2]
  double* Pforward;          // [n]
  DevMacHit* hits;           // [n]
  double Cshift;
  float mact;
  int32_t lds_cols;          // longest template of the launch (sizes the LDS sections)
  // forward / backward / DP run one launch per LENGTH CLASS of the batch (MacClasses): block b works on hit sel[b]
  const int32_t* sel;        // [n] hit numbers, grouped by class
  double* row_scratch;       // row state of the class whose templates do not fit into LDS: [block][10][lds_cols + 2]
  const int64_t* path_off;   // [n] capacity Lq+Lt+2 each
  int32_t* path_i;
  int32_t* path_j;
  signed char* path_state;
  float* path_S;
  float* path_P;
  const float* lg2;
  const float* diff;
  // secondary-structure scoring inside forward / backward (hit.ssm2 = 1 PRED_DSSP or 2 DSSP_PRED), null = none:
  // factor of cell (i, j) = ss_tab[mode-1][ss_qidx[mode-1][i] * (mode == 1 ? 8 : 44) + ss_tidx[col_off[k] + j]]
  const float* ss_tab;            // [2][352]  fpow2(ssw * S37[q_pred][q_conf][t_dssp]), fpow2(ssw * S73[q_dssp][t_pred][t_conf])
  const unsigned char* ss_qidx;   // [2][Lq+2]
  const unsigned char* ss_tidx;   // [cols + n]: per hit Lt+2 entries (index Lt+1 = what the reference reads past the template)
  const int64_t* ss_toff;         // [n] first entry of hit k in ss_tidx
  const int32_t* ss_mode;         // [n] 0, 1, 2
  // -o_matrices (hhv_mac_set_lists): dense planes laid out like mat, the value of the reference's sparse forward / backward
  // list entry where it has one, 0 elsewhere; null = not wanted
  float* fwd_list;
  float* bwd_list;
  uint32_t* err;             // the context's device error word (DEV_ERR_MAC_TIMEOUT), may be null
  // [n][Lq+2] (first, last) unmasked template column of every query row (hhv_mac_rowrange_kernel), first > last = none
  int2* row_rng;
  // hits whose template has this many columns or more run the dataflow kernels over the strips their rows' ranges need only
  // (hhv_mac.hip StripSpan); their F_MM / posterior planes are cleared by the kernel that leaves row_rng
  int32_t sparse_min_Lt;
  int32_t ring_min_Lt;       // ... and only templates of this many columns or more (shorter ones are in the class for capacity)
  int32_t ring_strips;       // single-wave kernels of the class without LDS: hits whose widest row (row_rng[0].x) is at most this many strips belong to the ring kernels (0 = none)
};
struct MacMaskArgs {
  const int4* ends;          // [n] i1, j1, i2, j2 of the Viterbi alignment
  const int64_t* vit_off;    // [n+1] Viterbi path steps (entries 1..nsteps of Hit::i / ::j, concatenated)
  const int32_t* vit_i;
  const int32_t* vit_j;
  const int64_t* excl_off;   // [n+1] cells of earlier MAC alignments of the same template (alt_i / alt_j)
  const int32_t* excl_i;
  const int32_t* excl_j;
  // resident hits (hhv_mac_realign_tset with hhv_mac_input::i == NULL): Viterbi alignment from the set's trace results
  const DevHit* res_hits;        // null = none
  const int32_t* res_template;   // [n] template index in the resident set, or -1 = path handed over by the host
  const int64_t* res_path_off;
  const int32_t* res_i;
  const int32_t* res_j;
  const int32_t* ranges;     // n_qranges query row ranges, then n_tranges template column ranges, (lo, hi) pairs
  int32_t n_qranges, n_tranges;
};
int launch_mac_mask(const MacArgs& a, const MacMaskArgs& m, void* stream);  // + MacArgs::row_rng
int launch_mac_rowrange(const MacArgs& a, void* stream);                     // MacArgs::row_rng of masks handed over by the host
// Length classes of a batch of hits.  A lone wave per hit keeps its row state (and, if it fits, the template) in LDS, and the
// LDS footprint of a launch is that of its longest template; one long template must not take the occupancy of the other
// hits nor set their limits, so the hits are launched class by class, every class on a stream of its own (they overlap):
//   0..3  template + row state + prefetch rows in LDS: up to 128 / 256 / 384 / ~800 columns (5 / 3 / 2 / 1 workgroups per CU)
//   4, 5  row state in LDS, template operands from global memory: up to 1022 / ~1450 columns (the dataflow kernels' LDS layout)
//   6     row state in global memory too (any length)
constexpr int MAC_CLASSES = 7;
struct MacClasses {
  int n[MAC_CLASSES];       // hits per class; MacArgs::sel lists class 0 first, then 1, ...
  int max_Lt[MAC_CLASSES];  // longest template per class
  int n_long;               // hits of the last class that are there by their length (the others: beyond the dataflow classes' budget)
};
// streams the class launches are spread over (launch_mac): s[0 .. MAC_CHAINS-1], fork / join events
constexpr int MAC_CHAINS = 7;  // (one per class: hhv_create asks the runtime for eight hardware queues)
struct MacStreams {
  void* s[MAC_CLASSES];
  void* fork;
  void* join[MAC_CLASSES];
};
int mac_length_class(int Lt, bool lds_allowed = true);
size_t mac_rows_lds(int max_Lt, bool stage);  // LDS of a forward / backward workgroup of a class whose longest template has max_Lt columns
void mac_dataflow_budget(int num_cus, int* max_hits, size_t* max_lds);  // what the dataflow classes 0 .. 5 of one batch may take
int launch_mac(const MacArgs& a, bool local, const MacClasses& cls, void* stream, const MacStreams* side);

// launchers implemented in hhv_kernels.hip
// W = lanes per systolic array (64, 32, 16): W < 64 variants exist for single-pass queries only (multi = false)
int launch_stream(int W, int R, bool local, bool bt, bool celloff, bool multi, bool ss, const StreamArgs& a, int n_waves, void* stream,
                  void* ev_start = nullptr, void* ev_stop = nullptr);
int stream_kernel_occupancy(int W, int R, bool local, bool bt, bool celloff, bool multi, bool ss, int* blocks_per_cu, int* vgprs);
int stream_kernel_waves(int W, bool ss);  // wavefronts per workgroup of the kernel launch_stream starts (8 for hhv_ss_kernel)
// per-W instantiation units (hhv_kernels.hip: 64, hhv_kernels_w32.hip, hhv_kernels_w16.hip)
void* stream_kernel_w64(int R, bool local, bool bt, bool celloff, bool multi, bool ss, bool first_strip);
// two-strip queries as one launch of two-wave workgroups (hhv_kernels_pair.hip)
int launch_pair(int R0, int R1, bool local, bool bt, int chain, bool ss, const StreamArgs& a, int n_pairs, void* stream, void* ev_start = nullptr,
                void* ev_stop = nullptr);
int pair_kernel_pairs_per_workgroup(bool ss);
int pair_kernel_occupancy(int R0, int R1, bool local, bool bt, int chain, bool ss);  // pairs (two-wave arrays) per CU, 0 = no such kernel
void* stream_kernel_w32(int R, bool local, bool bt, bool celloff, bool ss);
void* stream_kernel_w16(int R, bool local, bool bt, bool celloff, bool ss);
// one template's mask bytes -> cell-off entries; entries -> the reference's backtrace byte matrix (hhv_topk.hip)
int celloff_from_mask(uint64_t* bt, const int64_t* rec_off, const int32_t* L, int64_t pass_stride, int Lq, StripPlan plan, int t,
                      const unsigned char* d_mask /* (Lq+1) x (Lt+1) bytes, null = clear */, int Lt, hipStream_t stream);
int bt_matrix(const uint64_t* bt, const int64_t* rec_off, int64_t pass_stride, int Lq, StripPlan plan, int bt_mm, int t, int Lt,
              unsigned char* d_out /* (Lq+1) x (Lt+1) */, hipStream_t stream);
int launch_trace(const TraceArgs& a, void* stream);
int launch_pack_paths(const DevHit* hits, int n, const int64_t* path_off, const int32_t* pi, const int32_t* pj, const int8_t* ps,
                      const float* pS, const int64_t* out_off, uint16_t* oi, uint16_t* oj, int8_t* os, float* oS, void* stream);

}  // namespace hhv
