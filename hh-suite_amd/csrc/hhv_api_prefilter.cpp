// hhv_api_prefilter.cpp -- C ABI of the HHblits prefilter kernels.
#include "hhv_api_common.h"

using namespace hhv;
using hhv::api::dfree;
using hhv::api::fail;
using hhv::api::tset_init_common;

extern "C" {

// ---- HHblits prefilter kernels (N3) ----------------------------------------------------------------------
struct hhv_pfdb {
  hhv_ctx* ctx = nullptr;
  int32_t n = 0;
  int64_t total = 0;
  unsigned char* d_seqs = nullptr;
  unsigned char* d_carry[2] = {nullptr, nullptr};  // slab-to-slab diagonals of the gapless kernel (long queries), lazily
  size_t padded = 0;
  int64_t* d_off = nullptr;
  int32_t* d_order_all = nullptr;   // all sequences, longest first
  // per-call scratch, kept between calls (grown on demand): profile, scores, subset ids + their order, striped profile
  unsigned char* d_prof = nullptr;
  size_t prof_cap = 0;
  int32_t* d_scores = nullptr;
  int32_t* d_subset = nullptr;
  int32_t* d_order = nullptr;
  size_t jobs_cap = 0, sub_cap = 0, order_cap = 0;
  unsigned char* d_striped = nullptr;
  size_t striped_cap = 0;
  unsigned char* d_state = nullptr;  // H/E columns of the generic kernel for queries beyond LDS
  size_t state_cap = 0;
  // first selection step on the device: sort keys, sorted keys, radix-sort scratch, counter
  uint64_t* d_keys = nullptr;
  uint64_t* d_sorted = nullptr;
  void* d_sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  unsigned int* d_above = nullptr;
  std::vector<int32_t> length;      // host copy of the lengths
  int32_t max_len = 0;
};

// slots 0..n-1 ordered by descending sequence length (counting sort, stable): neighbouring jobs have similar
// lengths (the two halves of a wavefront finish together) and the long sequences start first
static void order_by_length(const std::vector<int32_t>& length, const int32_t* subset, int32_t n, int32_t max_len,
                            std::vector<int32_t>* order) {
  std::vector<int32_t> count((size_t)max_len + 2, 0);
  for (int k = 0; k < n; ++k) ++count[(size_t)max_len - length[subset ? subset[k] : k] + 1];
  for (size_t b = 1; b < count.size(); ++b) count[b] += count[b - 1];
  order->resize(n);
  for (int k = 0; k < n; ++k) (*order)[count[(size_t)max_len - length[subset ? subset[k] : k]]++] = k;
}

int hhv_prefilter_upload_db(hhv_ctx* c, int32_t n_db, const uint8_t* seqs, const int64_t* offsets, hhv_pfdb** out) {
  if (!c || !seqs || !offsets || !out || n_db < 1) return fail(HHV_E_ARG, "hhv_prefilter_upload_db: bad argument");
  *out = nullptr;
  if (offsets[0] != 0) return fail(HHV_E_ARG, "hhv_prefilter_upload_db: offsets[0] must be 0");
  for (int k = 0; k < n_db; ++k)
    if (offsets[k + 1] < offsets[k] || offsets[k + 1] - offsets[k] > (1 << 30))
      return fail(HHV_E_ARG, "hhv_prefilter_upload_db: bad offsets at %d", k);
  const int64_t total = offsets[n_db];
  for (int64_t b = 0; b < total; ++b)
    if (seqs[b] > 219) return fail(HHV_E_ARG, "hhv_prefilter_upload_db: state %d > 219 at byte %lld", seqs[b], (long long)b);
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_pfdb* db = new (std::nothrow) hhv_pfdb();
  if (!db) return fail(HHV_E_MEMORY, "out of host memory");
  db->ctx = c;
  db->n = n_db;
  db->total = total;
  db->length.resize(n_db);
  for (int k = 0; k < n_db; ++k) {
    db->length[k] = (int32_t)(offsets[k + 1] - offsets[k]);
    db->max_len = std::max(db->max_len, db->length[k]);
  }
  std::vector<int32_t> order;
  order_by_length(db->length, nullptr, n_db, db->max_len, &order);
  const size_t padded = ((size_t)total + 3) / 4 * 4 + 16;  // the kernels read whole aligned dwords
  db->padded = padded;
  if (hipMalloc(&db->d_seqs, padded) != hipSuccess || hipMalloc(&db->d_off, (size_t)(n_db + 1) * sizeof(int64_t)) != hipSuccess ||
      hipMalloc(&db->d_order_all, (size_t)n_db * sizeof(int32_t)) != hipSuccess ||
      hipMemset(db->d_seqs + ((size_t)total / 4 * 4), 0, padded - (size_t)total / 4 * 4) != hipSuccess ||
      hipMemcpy(db->d_seqs, seqs, (size_t)total, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db->d_off, offsets, (size_t)(n_db + 1) * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db->d_order_all, order.data(), (size_t)n_db * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
    hhv_prefilter_free_db(db);
    return fail(HHV_E_MEMORY, "hhv_prefilter_upload_db: device allocation/copy failed");
  }
  *out = db;
  return HHV_OK;
}

void hhv_prefilter_free_db(hhv_pfdb* db) {
  if (!db) return;
  if (db->ctx) (void)hipSetDevice(db->ctx->par.device);
  dfree(db->d_seqs);
  dfree(db->d_carry[0]);
  dfree(db->d_carry[1]);
  dfree(db->d_off);
  dfree(db->d_order_all);
  dfree(db->d_prof);
  dfree(db->d_scores);
  dfree(db->d_subset);
  dfree(db->d_order);
  dfree(db->d_striped);
  dfree(db->d_state);
  dfree(db->d_keys);
  dfree(db->d_sorted);
  dfree(db->d_sort_temp);
  dfree(db->d_above);
  delete db;
}

// runs the kernel; the scores stay in db->d_scores (and are copied to `scores` when it is not null)
static int prefilter_scores_impl(hhv_ctx* c, hhv_pfdb* db, const uint8_t* profile, int32_t Lq, int32_t score_offset,
                                 int32_t gapped, int32_t gap_init, int32_t gap_extend, const int32_t* subset, int32_t n_subset,
                                 int32_t* scores) {
  if (!c || !db || !profile) return fail(HHV_E_ARG, "hhv_prefilter_scores: null argument");
  if (db->ctx != c) return fail(HHV_E_ARG, "hhv_prefilter_scores: database belongs to another context");
  if (Lq < 1) return fail(HHV_E_ARG, "hhv_prefilter_scores: Lq = %d", Lq);
  if (score_offset < 0 || score_offset > 255 || gap_init < 0 || gap_extend < 0)
    return fail(HHV_E_ARG, "hhv_prefilter_scores: parameter out of range");
  const int64_t n_jobs = subset ? n_subset : db->n;
  if (n_jobs < 1) return HHV_OK;
  if (subset)
    for (int k = 0; k < n_subset; ++k)
      if (subset[k] < 0 || subset[k] >= db->n) return fail(HHV_E_ARG, "hhv_prefilter_scores: subset[%d] = %d", k, subset[k]);

  // kernel choice.  Fast kernels: profile as int8 (q - offset) in LDS, state in registers.
  const int W32 = (Lq + 31) / 32;  // 32 unsigned bytes per AVX2 vector of the reference (VECSIZE_INT * 4)
  // gapless: slabs of up to 512 query rows (W <= 8 cells per lane), any Lq; Smith-Waterman: W32 <= 20 (Lq <= 640)
  const int n_slabs = gapped ? 1 : (Lq + 511) / 512;
  const int slab_rows = gapped ? Lq : (Lq + n_slabs - 1) / n_slabs;
  const int Wfast = gapped ? W32 : (slab_rows + 63) / 64;
  bool fast = (gapped ? Lq <= 640 : true) && score_offset <= 128 && !getenv("HHV_PREFILTER_GENERIC");
  if (fast)
    for (size_t e = 0; e < (size_t)220 * Lq; ++e)
      if ((int)profile[e] - score_offset > 127) {
        fast = false;
        break;
      }
  const size_t state_lds = (size_t)8 * 3 * W32 * 32, prof_lds = (size_t)220 * W32 * 32;
  const bool generic_prof_lds = prof_lds + state_lds <= 160 * 1024;
  // the generic kernel keeps the H/E columns of its eight sequence slots in LDS up to Lq = 6816, beyond that in global memory
  const bool state_global = !fast && state_lds > 160 * 1024;
  const size_t lds = fast ? prefilter_fast_lds(gapped != 0, Wfast) : state_global ? 0 : state_lds + (generic_prof_lds ? prof_lds : 0);

  HIP_TRY(hipSetDevice(c->par.device));
  // scratch buffers live in the database handle: a search calls this twice per query
  auto grow = [](auto*& p, size_t& cap, size_t need_bytes) -> bool {
    if (cap >= need_bytes) return true;
    dfree(p);
    cap = 0;
    if (hipMalloc(&p, need_bytes) != hipSuccess) return false;
    cap = need_bytes;
    return true;
  };
  int rc = HHV_OK;
  std::vector<int32_t> order;
  unsigned char*& d_prof = db->d_prof;
  unsigned char*& d_striped = db->d_striped;
  int32_t*& d_subset = db->d_subset;
  int32_t*& d_order = db->d_order;
  int32_t*& d_scores = db->d_scores;
  std::vector<unsigned char> striped;
  if (subset) order_by_length(db->length, subset, n_subset, db->max_len, &order);
  if (!fast && !generic_prof_lds) {
    // Prefilter::stripe_query_profile layout (:386-425) for the kernel that reads the profile through L2
    striped.resize(prof_lds);
    for (int x = 0; x < 220; ++x)
      for (int j = 0; j < W32; ++j)
        for (int k = 0; k < 32; ++k) {
          const int p = k * W32 + j;
          striped[((size_t)x * W32 + j) * 32 + k] = p >= Lq ? (unsigned char)score_offset : profile[(size_t)x * Lq + p];
        }
  }
  if (!grow(db->d_prof, db->prof_cap, (size_t)220 * Lq) || !grow(db->d_scores, db->jobs_cap, (size_t)n_jobs * sizeof(int32_t)) ||
      (subset && (!grow(db->d_subset, db->sub_cap, (size_t)n_jobs * sizeof(int32_t)) ||
                  !grow(db->d_order, db->order_cap, (size_t)n_jobs * sizeof(int32_t)))) ||
      (!striped.empty() && !grow(db->d_striped, db->striped_cap, striped.size())) ||
      (state_global && !grow(db->d_state, db->state_cap, (size_t)c->num_cus * 8 * state_lds)))
    rc = fail(HHV_E_MEMORY, "hhv_prefilter_scores: device allocation failed");
  if (rc == HHV_OK &&
      (hipMemcpyAsync(d_prof, profile, (size_t)220 * Lq, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
       (subset && (hipMemcpyAsync(d_subset, subset, (size_t)n_jobs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                   hipMemcpyAsync(d_order, order.data(), (size_t)n_jobs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream) !=
                       hipSuccess)) ||
       (!striped.empty() && hipMemcpyAsync(d_striped, striped.data(), striped.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess)))
    rc = fail(HHV_E_DEVICE, "hhv_prefilter_scores: H2D copy failed");
  if (rc == HHV_OK) {
    PrefilterArgs a;
    a.profile = d_prof;
    a.striped = d_striped;
    a.seqs = db->d_seqs;
    a.offsets = db->d_off;
    a.subset = subset ? d_subset : nullptr;  // the scratch buffer outlives the call: only valid when this call filled it
    a.order = subset ? d_order : db->d_order_all;
    a.scores = d_scores;
    a.n_jobs = n_jobs;
    a.Lq = Lq;
    a.W = W32;
    a.offset = score_offset;
    a.gap_init = gap_init;
    a.gap_extend = gap_extend;
    a.q_base = 0;
    a.carry_in = nullptr;
    a.carry_out = nullptr;
    a.state_scratch = state_global ? db->d_state : nullptr;  // sized for num_cus * 8 blocks
    const int blocks_per_cu = std::max<int>(1, std::min<int>(fast ? 2 : 8, (int)((160 * 1024) / std::max<size_t>(lds, 1))));
    const int jobs_per_block = fast ? 16 : 8;
    const int n_blocks = (int)std::max<int64_t>(
        1, std::min<int64_t>((n_jobs + jobs_per_block - 1) / jobs_per_block, (int64_t)c->num_cus * blocks_per_cu));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    int lr = 0;
    if (fast && n_slabs > 1) {
      for (int b = 0; b < 2 && lr == 0; ++b)
        if (!db->d_carry[b] && hipMalloc(&db->d_carry[b], db->padded) != hipSuccess) lr = -(int)hipErrorOutOfMemory;
      for (int sl = 0; sl < n_slabs && lr == 0; ++sl) {
        a.q_base = sl * Wfast * 64;
        a.carry_in = sl ? db->d_carry[(sl - 1) & 1] : nullptr;
        a.carry_out = sl + 1 < n_slabs ? db->d_carry[sl & 1] : nullptr;
        lr = launch_prefilter_fast(a, false, Wfast, n_blocks, c->stream);
      }
    } else {
      lr = fast ? launch_prefilter_fast(a, gapped != 0, Wfast, n_blocks, c->stream)
                : launch_prefilter_generic(a, gapped != 0, generic_prof_lds, n_blocks, lds, c->stream);
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    if (lr != 0) rc = fail(HHV_E_DEVICE, "prefilter kernel launch failed: %s", hipGetErrorString((hipError_t)(-lr)));
  }
  if (rc == HHV_OK && scores &&
      (hipMemcpyAsync(scores, d_scores, (size_t)n_jobs * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
       hipStreamSynchronize(c->stream) != hipSuccess))
    rc = fail(HHV_E_DEVICE, "hhv_prefilter_scores: D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}

int hhv_prefilter_scores(hhv_ctx* c, hhv_pfdb* db, const uint8_t* profile, int32_t Lq, int32_t score_offset,
                         int32_t gapped, int32_t gap_init, int32_t gap_extend, const int32_t* subset, int32_t n_subset,
                         int32_t* scores) {
  if (!scores) return fail(HHV_E_ARG, "hhv_prefilter_scores: null argument");
  return prefilter_scores_impl(c, db, profile, Lq, score_offset, gapped, gap_init, gap_extend, subset, n_subset, scores);
}

int hhv_prefilter_first(hhv_ctx* c, hhv_pfdb* db, const uint8_t* profile, int32_t Lq, int32_t score_offset, float log_qlen,
                        int32_t bit_factor, int32_t smax_thresh, int32_t min_hits, int32_t* ids, int32_t cap, int32_t* n_out) {
  if (!ids || !n_out || cap < 0) return fail(HHV_E_ARG, "hhv_prefilter_first: bad argument");
  *n_out = 0;
  int rc = prefilter_scores_impl(c, db, profile, Lq, score_offset, 0, 0, 0, nullptr, 0, nullptr);
  if (rc != HHV_OK) return rc;
  const int n = db->n;
  if (!db->d_keys) {
    db->sort_temp_bytes = topk_temp_bytes(n);
    if (hipMalloc(&db->d_keys, (size_t)n * 8) != hipSuccess || hipMalloc(&db->d_sorted, (size_t)n * 8) != hipSuccess ||
        hipMalloc(&db->d_sort_temp, db->sort_temp_bytes) != hipSuccess || hipMalloc(&db->d_above, 16) != hipSuccess)
      return fail(HHV_E_MEMORY, "hhv_prefilter_first: device allocation failed");
  }
  HIP_TRY(hipMemsetAsync(db->d_above, 0, 4, c->stream));
  int lr = pf_select_sort(db->d_scores, db->d_off, n, log_qlen, bit_factor, smax_thresh, db->d_keys, db->d_sorted, db->d_sort_temp,
                          db->sort_temp_bytes, db->d_above, c->stream);
  if (lr != 0) return fail(HHV_E_DEVICE, "hhv_prefilter_first: sort failed: %s", hipGetErrorString((hipError_t)(-lr)));
  unsigned int above = 0;
  HIP_TRY(hipMemcpyAsync(&above, db->d_above, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // the reference keeps the min_hits best plus everything above the threshold (sorted: the first max(.,.) elements)
  const int m = (int)std::min<int64_t>(n, std::max<int64_t>(std::max(min_hits, 0), (int64_t)above));
  *n_out = m;
  if (m > cap) return fail(HHV_E_ARG, "hhv_prefilter_first: %d sequences pass, cap = %d", m, cap);
  if (m == 0) return HHV_OK;
  // ids through the (int32) subset scratch buffer
  if (db->sub_cap < (size_t)m * 4) {
    dfree(db->d_subset);
    db->sub_cap = 0;
    HIP_TRY(hipMalloc(&db->d_subset, (size_t)m * 4));
    db->sub_cap = (size_t)m * 4;
  }
  lr = pf_select_ids(db->d_sorted, m, db->d_subset, c->stream);
  if (lr != 0) return fail(HHV_E_DEVICE, "hhv_prefilter_first: gather failed");
  HIP_TRY(hipMemcpyAsync(ids, db->d_subset, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HHV_OK;
}


}  // extern "C"
