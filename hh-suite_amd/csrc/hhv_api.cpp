// hhv_api.cpp -- C-ABI host layer of libhhviterbi_hip.so (declared in include/hhviterbi_hip.h).
// Owns device memory, the packed template sets, the wave partition of the template stream and the
// kernel launches.  There is deliberately no CPU compute path here: every entry point that needs
// arithmetic launches a HIP kernel and fails with HHV_E_DEVICE when no device is usable.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "../../include/hhviterbi_hip.h"
#include "hhv_internal.h"
#include "hhv_pack.h"
#include "viterbi_lane.h"

using namespace hhv;

static_assert(sizeof(hhv_result) == sizeof(DevResult), "hhv_result layout");
static_assert(sizeof(hhv_hit) == sizeof(DevHit), "hhv_hit layout");
static_assert(HHV_STREAM_PAD == STREAM_PAD_RECS, "stream pad");

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorOutOfMemory ? HHV_E_MEMORY : HHV_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
void dfree(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

}  // namespace

struct hhv_ctx {
  hhv_params par;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  int num_cus = 0;
  // query: P passes of 64*R rows each (P = 1 up to Lq = 320)
  int Lq = 0, R = 0, P = 0;
  float* d_qpack = nullptr;  // [P*64*R][28]
  float* d_qp = nullptr;     // [(Lq+1)][20] AoS, for the backtrace rescoring
  // fast_log2 tables (src/util-inl.h:108-130)
  float* d_lg2 = nullptr;
  float* d_diff = nullptr;
  // secondary structure
  std::vector<float> S73, S33, S37;                    // host copies of the score tables
  std::vector<int8_t> q_pred, q_conf, q_dssp;          // [Lq+1], empty = absent
  int ss_hmm_mode = 0;                                 // HMM::NO_SS_INFORMATION
  bool ss_dirty = true;
  void* mac_cache = nullptr;                           // one recycled device block of the MAC realignment
  size_t mac_cache_bytes = 0;
  float* d_ss_table = nullptr;                         // ssw * table of the current mode
  int32_t* d_ss_q_off = nullptr;                       // [P*64*R]
  int ss_t_shift = 0, ss_t_mask = 0;
};

struct hhv_tset {
  hhv_ctx* ctx = nullptr;
  int32_t n = 0;
  std::vector<int32_t> L;
  std::vector<int64_t> rec_off;  // [n+1]: header record of template k; rec_off[n] = terminal header
  int64_t n_records = 0;         // rec_off[n] + 1
  float* d_records = nullptr;
  bool owns_records = false;
  int64_t* d_rec_off = nullptr;
  int32_t* d_L = nullptr;
  DevResult* d_results = nullptr;
  // wave partition
  int n_waves = 0;
  int64_t* d_wave_rec = nullptr;
  // backtrace bytes: [pass][record][lane] entries
  uint64_t* d_bt = nullptr;
  bool bt_valid = false;
  int bt_Lq = 0, bt_R = 0, bt_P = 0;
  // carry between the passes of a long query
  float4* d_carry = nullptr;
  float* d_carry_mi = nullptr;
  // trace outputs
  std::vector<int64_t> path_off;
  int path_Lq = -1;
  int64_t* d_path_off = nullptr;
  int32_t* d_i_steps = nullptr;
  int32_t* d_j_steps = nullptr;
  int8_t* d_states = nullptr;
  float* d_S = nullptr;
  DevHit* d_hits = nullptr;
  bool hits_valid = false;
  // top-k scratch
  DevHit* d_topk = nullptr;
  int topk_cap = 0;
  uint64_t* d_keys = nullptr;
  uint64_t* d_sorted = nullptr;
  void* d_sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  DevHit* d_raw_hits = nullptr;
};

struct hhv_rawset {
  hhv_ctx* ctx = nullptr;
  int32_t n = 0;
  std::vector<int32_t> L;
  std::vector<int64_t> rec_off;
  int64_t n_cols = 0;
  float* d_raw = nullptr;
  float* d_neff_hmm = nullptr;
  float* d_p_tmp = nullptr;
  float* d_tr_tmp = nullptr;
  float* d_pav = nullptr;
  float* d_pb = nullptr;
  float* d_R = nullptr;
  float* d_qpav = nullptr;
  // length classes of the prepare kernels: fused with small LDS (L <= 447), fused with large LDS (L <= 1300), split
  int32_t* d_ids[3] = {nullptr, nullptr, nullptr};
  int32_t n_ids[3] = {0, 0, 0};
  int32_t max_L[3] = {0, 0, 0};
  bool prepared = false;
};

namespace hhv {
size_t topk_temp_bytes(int n);
void results_to_hits(const DevResult* d_res, int n, DevHit* d_hits, hipStream_t stream);
int topk_device(const DevHit* d_hits, int n, int k, DevHit* d_out, uint64_t* keys, uint64_t* sorted, void* temp,
                size_t temp_bytes, hipStream_t stream, std::string* err);
}

extern "C" {

int hhv_abi_version(void) { return HHV_ABI_VERSION; }
const char* hhv_last_error(void) { return g_err.c_str(); }

int32_t hhv_record_bytes(void) { return REC_DW * (int32_t)sizeof(float); }

int hhv_pack_profile(const float* p, const float* tr, int32_t L, int32_t index, float* out) {
  if (!p || !tr || !out || L < 1) return fail(HHV_E_ARG, "hhv_pack_profile: bad argument");
  if (index >= 0) pack_template(p, tr, L, index, out);
  else pack_columns(p, tr, L, out);
  return HHV_OK;
}

// src/util-inl.h:108-130: lg2[i] = log(float(1024+i))*1.442695041 - 10.0f, diff[i-1] = (lg2[i]-prev)*1.2352E-4.
// The reference initialises this table in whichever translation unit calls fast_log2 first; in a real run that
// is hhhmm.cpp (HMM::AddTransitionPseudocounts during PrepareQueryHMM), where log(float) binds to the DOUBLE
// log - not hhviterbi.cpp, where it would bind to logf.  The double flavour is therefore the one the Viterbi
// rescoring sees in hhsearch/hhblits, and the one built here (pinned through the oracle against oracle/_ref).
int hhv_fast_log2_tables(float* lg2, float* diff) {
  if (!lg2 || !diff) return fail(HHV_E_ARG, "hhv_fast_log2_tables: null");
  float prev = 0.0f;
  lg2[0] = 0.0f;
  diff[1024] = 0.0f;
  for (int i = 1; i <= 1024; ++i) {
    lg2[i] = (float)(log((double)(float)(1024 + i)) * 1.442695041 - (double)10.0f);
    diff[i - 1] = (float)((double)(lg2[i] - prev) * 1.2352E-4);
    prev = lg2[i];
  }
  return HHV_OK;
}

int hhv_create(hhv_ctx** out, const hhv_params* par) {
  if (!out || !par) return fail(HHV_E_ARG, "hhv_create: null argument");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(HHV_E_DEVICE, "hhv_create: no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  if (par->device < 0 || par->device >= ndev) return fail(HHV_E_ARG, "hhv_create: device %d of %d", par->device, ndev);
  HIP_TRY(hipSetDevice(par->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, par->device));
  hhv_ctx* c = new (std::nothrow) hhv_ctx();
  if (!c) return fail(HHV_E_MEMORY, "hhv_create: out of host memory");
  c->par = *par;
  c->num_cus = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    hhv_destroy(c);
    return fail(HHV_E_DEVICE, "hhv_create: stream/event creation failed");
  }
  std::vector<float> lg2(1025), diff(1025);
  hhv_fast_log2_tables(lg2.data(), diff.data());
  if (hipMalloc(&c->d_lg2, 1025 * sizeof(float)) != hipSuccess ||
      hipMalloc(&c->d_diff, 1025 * sizeof(float)) != hipSuccess ||
      hipMemcpy(c->d_lg2, lg2.data(), 1025 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->d_diff, diff.data(), 1025 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    hhv_destroy(c);
    return fail(HHV_E_DEVICE, "hhv_create: table upload failed");
  }
  *out = c;
  return HHV_OK;
}

void hhv_destroy(hhv_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->par.device);
  dfree(c->d_qpack);
  dfree(c->d_qp);
  dfree(c->d_lg2);
  dfree(c->d_diff);
  dfree(c->d_ss_table);
  dfree(c->d_ss_q_off);
  dfree(c->mac_cache);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int hhv_set_query(hhv_ctx* c, const float* p, const float* tr, int32_t Lq) {
  if (!c || !p || !tr) return fail(HHV_E_ARG, "hhv_set_query: null argument");
  if (Lq < 1) return fail(HHV_E_ARG, "hhv_set_query: Lq = %d", Lq);
  if (Lq > 0x7FFF) return fail(HHV_E_LIMIT, "hhv_set_query: Lq = %d exceeds 32767", Lq);
  // strips of 64*R rows (R <= 5 keeps the kernel at 2 waves/SIMD).  Cost of a pass per stream record ~ a fixed
  // per-step overhead (LDS read, DPP hand-off, loop) of ~0.7 cell-equivalents plus R cells: pick the (R, P)
  // minimising P * (0.7 + R), ties -> fewer passes.
  int R = 1, P = 0;
  {
    double best = -1.0;
    for (int r = 1; r <= MAX_R; ++r) {
      const int p = (Lq + LANES * r - 1) / (LANES * r);
      const double cost = p * (0.7 + r);
      if (best < 0 || cost < best - 1e-9 || (cost < best + 1e-9 && p < P)) {
        best = cost;
        R = r;
        P = p;
      }
    }
  }
  HIP_TRY(hipSetDevice(c->par.device));
  std::vector<float> qpack((size_t)P * LANES * R * REC_DW, 0.0f);
  pack_columns(p, tr, Lq, qpack.data());
  dfree(c->d_qpack);
  dfree(c->d_qp);
  HIP_TRY(hipMalloc(&c->d_qpack, qpack.size() * sizeof(float)));
  HIP_TRY(hipMalloc(&c->d_qp, (size_t)(Lq + 1) * 20 * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(c->d_qpack, qpack.data(), qpack.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_qp, p, (size_t)(Lq + 1) * 20 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->Lq = Lq;
  c->R = R;
  c->P = P;
  c->q_pred.clear();
  c->q_conf.clear();
  c->q_dssp.clear();
  c->ss_dirty = true;
  return HHV_OK;
}

int hhv_set_ss_tables(hhv_ctx* c, const float* S73, const float* S33, const float* S37) {
  if (!c || !S73 || !S33 || !S37) return fail(HHV_E_ARG, "hhv_set_ss_tables: null argument");
  c->S73.assign(S73, S73 + 8 * 4 * 11);
  c->S33.assign(S33, S33 + 4 * 11 * 4 * 11);
  c->S37.assign(S37, S37 + 4 * 11 * 8);
  c->ss_dirty = true;
  return HHV_OK;
}

int hhv_set_query_ss(hhv_ctx* c, const int8_t* ss_pred, const int8_t* ss_conf, const int8_t* ss_dssp) {
  if (!c) return fail(HHV_E_ARG, "hhv_set_query_ss: null argument");
  if (c->Lq < 1) return fail(HHV_E_STATE, "hhv_set_query_ss: call hhv_set_query first");
  const size_t m = (size_t)c->Lq + 1;
  c->q_pred.clear();
  c->q_conf.clear();
  c->q_dssp.clear();
  if (ss_pred) c->q_pred.assign(ss_pred, ss_pred + m);
  if (ss_conf) c->q_conf.assign(ss_conf, ss_conf + m);
  if (ss_dssp) c->q_dssp.assign(ss_dssp, ss_dssp + m);
  c->ss_dirty = true;
  return HHV_OK;
}

int hhv_set_ss_mode(hhv_ctx* c, int32_t mode) {
  if (!c) return fail(HHV_E_ARG, "hhv_set_ss_mode: null argument");
  if (mode != 0 && mode != 1 && mode != 2 && mode != 4)
    return fail(HHV_E_ARG, "hhv_set_ss_mode: %d is not one of 0 (none), 1 (PRED_DSSP), 2 (DSSP_PRED), 4 (PRED_PRED)", mode);
  if (mode != 0 && c->S33.empty()) return fail(HHV_E_STATE, "hhv_set_ss_mode: call hhv_set_ss_tables first");
  c->ss_hmm_mode = mode;
  c->ss_dirty = true;
  return HHV_OK;
}

// (Re)build the device-side SS operands of the current mode: the premultiplied table ssw*S (the reference
// multiplies per cell, src/hhviterbialgorithm.cpp:209 - the same single fp32 product) and the per-row offsets
//   PRED_PRED: S33[q_pred][q_conf][t_pred][t_conf]  row (q_pred*11+q_conf)*44, column pred_index   (:199-200)
//   DSSP_PRED: S73[q_dssp][t_pred][t_conf]          row q_dssp*44,              column pred_index   (:201-202)
//   PRED_DSSP: S37[q_pred][q_conf][t_dssp]          row (q_pred*11+q_conf)*8,   column dssp_index   (:203-204)
static int ensure_ss(hhv_ctx* c) {
  if (!c->ss_dirty) return HHV_OK;
  dfree(c->d_ss_table);
  dfree(c->d_ss_q_off);
  c->ss_dirty = false;
  if (c->ss_hmm_mode == 0 || c->Lq < 1) return HHV_OK;
  const std::vector<float>& T = c->ss_hmm_mode == 4 ? c->S33 : (c->ss_hmm_mode == 2 ? c->S73 : c->S37);
  std::vector<float> tab(T.size());
  for (size_t k = 0; k < T.size(); ++k) tab[k] = c->par.ssw * T[k];
  const size_t rows = (size_t)c->P * LANES * c->R;
  std::vector<int32_t> off(rows, 0);
  for (int i = 1; i <= c->Lq; ++i) {
    const int pred = c->q_pred.empty() ? 0 : (unsigned char)c->q_pred[i], conf = c->q_conf.empty() ? 0 : c->q_conf[i];
    const int dssp = c->q_dssp.empty() ? 0 : (unsigned char)c->q_dssp[i];
    int o;
    if (c->ss_hmm_mode == 4) o = (pred * 11 + conf) * 44;
    else if (c->ss_hmm_mode == 2) o = dssp * 44;
    else o = (pred * 11 + conf) * 8;
    if (o < 0 || (size_t)o + (c->ss_hmm_mode == 1 ? 8 : 44) > T.size())
      return fail(HHV_E_ARG, "query secondary-structure code out of range at row %d", i);
    off[i - 1] = o;
  }
  c->ss_t_shift = c->ss_hmm_mode == 1 ? META_DSSP_SHIFT : META_PRED_SHIFT;
  c->ss_t_mask = c->ss_hmm_mode == 1 ? META_DSSP_MASK : META_PRED_MASK;
  HIP_TRY(hipMalloc(&c->d_ss_table, tab.size() * sizeof(float)));
  HIP_TRY(hipMalloc(&c->d_ss_q_off, off.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(c->d_ss_table, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_ss_q_off, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return HHV_OK;
}

static int tset_init_common(hhv_ctx* c, hhv_tset* ts, int32_t n, const int32_t* L) {
  ts->ctx = c;
  ts->n = n;
  ts->L.assign(L, L + n);
  ts->rec_off.resize((size_t)n + 1);
  int64_t off = 0;
  for (int k = 0; k < n; ++k) {
    if (L[k] < 1 || L[k] > 0xFFFF) return fail(HHV_E_ARG, "template %d: L = %d out of range [1, 65535]", k, L[k]);
    ts->rec_off[k] = off;
    off += (int64_t)L[k] + 1;
  }
  ts->rec_off[n] = off;
  ts->n_records = off + 1;
  HIP_TRY(hipMalloc(&ts->d_rec_off, (size_t)(n + 1) * sizeof(int64_t)));
  HIP_TRY(hipMalloc(&ts->d_L, (size_t)n * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&ts->d_results, (size_t)n * sizeof(DevResult)));
  HIP_TRY(hipMemcpy(ts->d_rec_off, ts->rec_off.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ts->d_L, ts->L.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
  return HHV_OK;
}

int hhv_upload_templates(hhv_ctx* c, int32_t n, const int32_t* L, const float* const* p, const float* const* tr,
                         hhv_tset** out) {
  return hhv_upload_templates_ss(c, n, L, p, tr, nullptr, nullptr, nullptr, out);
}

int hhv_upload_templates_ss(hhv_ctx* c, int32_t n, const int32_t* L, const float* const* p, const float* const* tr,
                            const int8_t* const* ss_pred, const int8_t* const* ss_conf, const int8_t* const* ss_dssp,
                            hhv_tset** out) {
  if (!c || !L || !p || !tr || !out) return fail(HHV_E_ARG, "hhv_upload_templates: null argument");
  if (n < 1) return fail(HHV_E_ARG, "hhv_upload_templates: n = %d", n);
  *out = nullptr;
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_tset* ts = new (std::nothrow) hhv_tset();
  if (!ts) return fail(HHV_E_MEMORY, "out of host memory");
  int rc = tset_init_common(c, ts, n, L);
  if (rc != HHV_OK) {
    hhv_tset_free(ts);
    return rc;
  }
  const size_t total = (size_t)(ts->n_records + STREAM_PAD_RECS) * REC_DW;
  if (hipMalloc(&ts->d_records, total * sizeof(float)) != hipSuccess) {
    hhv_tset_free(ts);
    return fail(HHV_E_MEMORY, "hhv_upload_templates: device allocation of %zu bytes failed", total * sizeof(float));
  }
  ts->owns_records = true;
  // pack and upload in slabs of <= 64 MiB of pinned-size staging
  const size_t slab_recs = (64u << 20) / (REC_DW * sizeof(float));
  std::vector<float> stage;
  int k = 0;
  while (k < n) {
    const int k0 = k;
    size_t recs = 0;
    while (k < n && (recs == 0 || recs + (size_t)L[k] + 1 <= slab_recs)) {
      recs += (size_t)L[k] + 1;
      ++k;
    }
    stage.resize(recs * REC_DW);
    size_t o = 0;
    for (int t = k0; t < k; ++t) {
      if (!p[t] || !tr[t]) {
        hhv_tset_free(ts);
        return fail(HHV_E_ARG, "hhv_upload_templates: template %d has a null profile", t);
      }
      for (int j = 1; j <= L[t]; ++j) {
        const int pr = ss_pred && ss_pred[t] ? ss_pred[t][j] : 0, cf = ss_conf && ss_conf[t] ? ss_conf[t][j] : 0;
        const int ds = ss_dssp && ss_dssp[t] ? ss_dssp[t][j] : 0;
        if (pr < 0 || pr > 3 || cf < 0 || cf > 10 || ds < 0 || ds > 7) {
          hhv_tset_free(ts);
          return fail(HHV_E_ARG, "template %d column %d: secondary-structure code out of range", t, j);
        }
      }
      pack_template(p[t], tr[t], L[t], t, stage.data() + o, ss_pred ? ss_pred[t] : nullptr,
                    ss_conf ? ss_conf[t] : nullptr, ss_dssp ? ss_dssp[t] : nullptr);
      o += ((size_t)L[t] + 1) * REC_DW;
    }
    if (hipMemcpy(ts->d_records + (size_t)ts->rec_off[k0] * REC_DW, stage.data(), stage.size() * sizeof(float),
                  hipMemcpyHostToDevice) != hipSuccess) {
      hhv_tset_free(ts);
      return fail(HHV_E_DEVICE, "hhv_upload_templates: H2D copy failed");
    }
  }
  // terminal header + zeroed slack
  std::vector<float> tail((size_t)(1 + STREAM_PAD_RECS) * REC_DW, 0.0f);
  write_header(tail.data(), -1, 0);
  if (hipMemcpy(ts->d_records + (size_t)ts->rec_off[n] * REC_DW, tail.data(), tail.size() * sizeof(float),
                hipMemcpyHostToDevice) != hipSuccess) {
    hhv_tset_free(ts);
    return fail(HHV_E_DEVICE, "hhv_upload_templates: H2D copy failed");
  }
  *out = ts;
  return HHV_OK;
}

int hhv_adopt_device_stream(hhv_ctx* c, int32_t n, const int32_t* L, const void* d_records, hhv_tset** out) {
  if (!c || !L || !d_records || !out) return fail(HHV_E_ARG, "hhv_adopt_device_stream: null argument");
  if (n < 1) return fail(HHV_E_ARG, "hhv_adopt_device_stream: n = %d", n);
  *out = nullptr;
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_tset* ts = new (std::nothrow) hhv_tset();
  if (!ts) return fail(HHV_E_MEMORY, "out of host memory");
  int rc = tset_init_common(c, ts, n, L);
  if (rc != HHV_OK) {
    hhv_tset_free(ts);
    return rc;
  }
  ts->d_records = (float*)d_records;
  ts->owns_records = false;
  *out = ts;
  return HHV_OK;
}

// ---- on-device PrepareTemplateHMM (N2) ---------------------------------------------------------------
void hhv_rawset_free(hhv_rawset* rs);
// raw HMMs -> the 32-dword raw column block the prepare kernels read (hhv_internal.h RAW_*)
static int build_raw_block(int32_t n, const int32_t* L, const float* const* f, const float* const* tr, const float* const* neff,
                           const int8_t* const* ss_pred, const int8_t* const* ss_conf, const int8_t* const* ss_dssp,
                           std::vector<float>* host) {
  int64_t off = 0;
  for (int k = 0; k < n; ++k) {
    if (L[k] < 1 || L[k] > 0xFFFF || !f[k] || !tr[k] || !neff[k]) return fail(HHV_E_ARG, "raw template %d invalid", k);
    off += (int64_t)L[k] + 1;
  }
  host->assign((size_t)off * RAW_DW, 0.0f);
  off = 0;
  for (int k = 0; k < n; ++k) {
    for (int i = 0; i <= L[k]; ++i) {
      float* w = host->data() + (size_t)(off + i) * RAW_DW;
      memcpy(w + RAW_F, f[k] + (size_t)i * 20, 20 * sizeof(float));
      memcpy(w + RAW_TR, tr[k] + (size_t)i * 7, 7 * sizeof(float));
      memcpy(w + RAW_NEFF, neff[k] + (size_t)i * 3, 3 * sizeof(float));
      int32_t meta = i;
      if (i >= 1) {
        const int pr = ss_pred && ss_pred[k] ? (unsigned char)ss_pred[k][i] : 0, cf = ss_conf && ss_conf[k] ? ss_conf[k][i] : 0;
        const int ds = ss_dssp && ss_dssp[k] ? (unsigned char)ss_dssp[k][i] : 0;
        meta |= (int32_t)((unsigned char)(pr * 11 + cf) & META_PRED_MASK) << META_PRED_SHIFT;
        meta |= (int32_t)(ds & META_DSSP_MASK) << META_DSSP_SHIFT;
      }
      memcpy(w + RAW_J, &meta, 4);
      const int32_t Lk = L[k];
      memcpy(w + RAW_L, &Lk, 4);
    }
    off += (int64_t)L[k] + 1;
  }
  return HHV_OK;
}

// raw column block -> resident raw set (the block may come from build_raw_block or straight from a raw database file)
static int rawset_from_block(hhv_ctx* c, int32_t n, const int32_t* L, const float* neff_hmm, const float* block,
                             size_t block_floats, hhv_rawset** out) {
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_rawset* rs = new (std::nothrow) hhv_rawset();
  if (!rs) return fail(HHV_E_MEMORY, "out of host memory");
  rs->ctx = c;
  rs->n = n;
  rs->L.assign(L, L + n);
  rs->rec_off.resize((size_t)n + 1);
  int64_t off = 0;
  for (int k = 0; k < n; ++k) {
    if (L[k] < 1 || L[k] > 0xFFFF) {
      delete rs;
      return fail(HHV_E_ARG, "raw template %d: length %d", k, L[k]);
    }
    rs->rec_off[k] = off;
    off += (int64_t)L[k] + 1;
  }
  rs->rec_off[n] = off;
  rs->n_cols = off;
  if ((size_t)off * RAW_DW != block_floats) {
    delete rs;
    return fail(HHV_E_ARG, "raw column block has %zu floats, expected %zu", block_floats, (size_t)off * RAW_DW);
  }
  // length classes of the prepare kernels (hhv_prep.hip): the fused kernel keeps a template in LDS
  std::vector<int32_t> cls_ids[3];
  for (int k = 0; k < n; ++k) {
    const int cls = L[k] <= 447 ? 0 : (L[k] <= 1300 ? 1 : 2);
    cls_ids[cls].push_back(k);
    rs->max_L[cls] = std::max(rs->max_L[cls], L[k]);
  }
  bool ok = true;
  for (int cls = 0; cls < 3 && ok; ++cls) {
    rs->n_ids[cls] = (int32_t)cls_ids[cls].size();
    if (rs->n_ids[cls] == 0) continue;
    ok = hipMalloc(&rs->d_ids[cls], cls_ids[cls].size() * sizeof(int32_t)) == hipSuccess &&
         hipMemcpy(rs->d_ids[cls], cls_ids[cls].data(), cls_ids[cls].size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  }
  // the intermediate of the split path (columns indexed like the raw stream) exists only if a template needs it
  if (ok && rs->n_ids[2] > 0)
    ok = hipMalloc(&rs->d_p_tmp, (size_t)off * 20 * sizeof(float)) == hipSuccess &&
         hipMalloc(&rs->d_tr_tmp, (size_t)off * 8 * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc(&rs->d_raw, block_floats * sizeof(float)) == hipSuccess &&
            hipMalloc(&rs->d_neff_hmm, (size_t)n * sizeof(float)) == hipSuccess &&
            hipMalloc(&rs->d_pav, (size_t)n * 20 * sizeof(float)) == hipSuccess &&
            hipMalloc(&rs->d_pb, 20 * sizeof(float)) == hipSuccess && hipMalloc(&rs->d_R, 400 * sizeof(float)) == hipSuccess &&
            hipMalloc(&rs->d_qpav, 20 * sizeof(float)) == hipSuccess &&
            hipMemcpy(rs->d_raw, block, block_floats * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(rs->d_neff_hmm, neff_hmm, (size_t)n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    hhv_rawset_free(rs);
    return fail(HHV_E_MEMORY, "raw template set: device allocation/copy failed");
  }
  *out = rs;
  return HHV_OK;
}

int hhv_upload_raw_templates(hhv_ctx* c, int32_t n, const int32_t* L, const float* const* f, const float* const* tr,
                             const float* const* neff, const float* neff_hmm, const int8_t* const* ss_pred,
                             const int8_t* const* ss_conf, const int8_t* const* ss_dssp, hhv_rawset** out) {
  if (!c || !L || !f || !tr || !neff || !neff_hmm || !out) return fail(HHV_E_ARG, "hhv_upload_raw_templates: null argument");
  if (n < 1) return fail(HHV_E_ARG, "hhv_upload_raw_templates: n = %d", n);
  *out = nullptr;
  std::vector<float> host;
  const int rc = build_raw_block(n, L, f, tr, neff, ss_pred, ss_conf, ss_dssp, &host);
  if (rc != HHV_OK) return rc;
  return rawset_from_block(c, n, L, neff_hmm, host.data(), host.size(), out);
}

// Raw template database file (N1 for the N2 path): header, lengths, Neff_HMM, then the raw column block exactly as it
// sits in HBM - built once from the .hhm files, loaded per search without parsing or repacking.
namespace {
struct RawDbHeader {
  char magic[8];
  int32_t n;
  int32_t column_dwords;
  int64_t n_cols;
  char pad[40];
};
static_assert(sizeof(RawDbHeader) == 64, "raw db header");
}  // namespace

int hhv_rawdb_write(const char* path, int32_t n, const int32_t* L, const float* const* f, const float* const* tr,
                    const float* const* neff, const float* neff_hmm, const int8_t* const* ss_pred,
                    const int8_t* const* ss_conf, const int8_t* const* ss_dssp) {
  if (!path || !L || !f || !tr || !neff || !neff_hmm || n < 1) return fail(HHV_E_ARG, "hhv_rawdb_write: bad argument");
  std::vector<float> host;
  const int rc = build_raw_block(n, L, f, tr, neff, ss_pred, ss_conf, ss_dssp, &host);
  if (rc != HHV_OK) return rc;
  FILE* fp = fopen(path, "wb");
  if (!fp) return fail(HHV_E_ARG, "hhv_rawdb_write: cannot open %s", path);
  RawDbHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "HHVRAW01", 8);
  h.n = n;
  h.column_dwords = RAW_DW;
  h.n_cols = (int64_t)(host.size() / RAW_DW);
  bool ok = fwrite(&h, sizeof(h), 1, fp) == 1 && fwrite(L, sizeof(int32_t), (size_t)n, fp) == (size_t)n &&
            fwrite(neff_hmm, sizeof(float), (size_t)n, fp) == (size_t)n &&
            fwrite(host.data(), sizeof(float), host.size(), fp) == host.size();
  ok = (fclose(fp) == 0) && ok;
  return ok ? HHV_OK : fail(HHV_E_ARG, "hhv_rawdb_write: write to %s failed", path);
}

int hhv_rawdb_open(hhv_ctx* c, const char* path, hhv_rawset** out) {
  if (!c || !path || !out) return fail(HHV_E_ARG, "hhv_rawdb_open: null argument");
  *out = nullptr;
  FILE* fp = fopen(path, "rb");
  if (!fp) return fail(HHV_E_ARG, "hhv_rawdb_open: cannot open %s", path);
  RawDbHeader h;
  if (fread(&h, sizeof(h), 1, fp) != 1 || memcmp(h.magic, "HHVRAW01", 8) != 0 || h.column_dwords != RAW_DW || h.n < 1 ||
      h.n_cols < 2) {
    fclose(fp);
    return fail(HHV_E_ARG, "hhv_rawdb_open: %s is not a raw template database", path);
  }
  std::vector<int32_t> L((size_t)h.n);
  std::vector<float> neff_hmm((size_t)h.n), block((size_t)h.n_cols * RAW_DW);
  const bool ok = fread(L.data(), sizeof(int32_t), L.size(), fp) == L.size() &&
                  fread(neff_hmm.data(), sizeof(float), neff_hmm.size(), fp) == neff_hmm.size() &&
                  fread(block.data(), sizeof(float), block.size(), fp) == block.size();
  fclose(fp);
  if (!ok) return fail(HHV_E_ARG, "hhv_rawdb_open: %s is truncated", path);
  return rawset_from_block(c, h.n, L.data(), neff_hmm.data(), block.data(), block.size(), out);
}

int32_t hhv_rawset_size(const hhv_rawset* rs) { return rs ? rs->n : 0; }
int hhv_rawset_lengths(const hhv_rawset* rs, int32_t* L) {
  if (!rs || !L) return fail(HHV_E_ARG, "hhv_rawset_lengths: null argument");
  memcpy(L, rs->L.data(), (size_t)rs->n * sizeof(int32_t));
  return HHV_OK;
}

void hhv_rawset_free(hhv_rawset* rs) {
  if (!rs) return;
  if (rs->ctx) (void)hipSetDevice(rs->ctx->par.device);
  dfree(rs->d_raw);
  dfree(rs->d_neff_hmm);
  dfree(rs->d_p_tmp);
  dfree(rs->d_tr_tmp);
  dfree(rs->d_pav);
  dfree(rs->d_pb);
  dfree(rs->d_R);
  dfree(rs->d_qpav);
  for (int cls = 0; cls < 3; ++cls) dfree(rs->d_ids[cls]);
  delete rs;
}

int hhv_prepare_templates(hhv_ctx* c, hhv_rawset* rs, const hhv_prep_params* par, const float* q_pav, hhv_tset** out) {
  if (!c || !rs || !par || !q_pav || !out) return fail(HHV_E_ARG, "hhv_prepare_templates: null argument");
  if (rs->ctx != c) return fail(HHV_E_ARG, "hhv_prepare_templates: raw set belongs to another context");
  if (par->pcm < 0 || par->pcm > 2) return fail(HHV_E_LIMIT, "hhv_prepare_templates: pcm = %d (only 0, 1, 2)", par->pcm);
  if (par->pcm == 2 && par->pcc != 1.0f)
    return fail(HHV_E_LIMIT, "hhv_prepare_templates: pcc = %g; only the default pcc = 1 avoids libm pow() and is built", par->pcc);
  if (par->columnscore < 0 || par->columnscore > 3)
    return fail(HHV_E_LIMIT, "hhv_prepare_templates: columnscore = %d (only 0..3)", par->columnscore);
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_tset* ts = *out;
  if (!ts) {
    ts = new (std::nothrow) hhv_tset();
    if (!ts) return fail(HHV_E_MEMORY, "out of host memory");
    int rc = tset_init_common(c, ts, rs->n, rs->L.data());
    if (rc == HHV_OK && hipMalloc(&ts->d_records, (size_t)(ts->n_records + STREAM_PAD_RECS) * REC_DW * sizeof(float)) != hipSuccess)
      rc = fail(HHV_E_MEMORY, "hhv_prepare_templates: device allocation failed");
    if (rc != HHV_OK) {
      hhv_tset_free(ts);
      return rc;
    }
    ts->owns_records = true;
    std::vector<float> tail((size_t)(1 + STREAM_PAD_RECS) * REC_DW, 0.0f);
    write_header(tail.data(), -1, 0);
    HIP_TRY(hipMemcpy(ts->d_records + (size_t)ts->rec_off[rs->n] * REC_DW, tail.data(), tail.size() * sizeof(float),
                      hipMemcpyHostToDevice));
  } else if (ts->n != rs->n || ts->n_records != rs->n_cols + 1) {
    return fail(HHV_E_ARG, "hhv_prepare_templates: *out was not created from this raw set");
  }
  HIP_TRY(hipMemcpyAsync(rs->d_pb, par->pb, 20 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(rs->d_R, par->R, 400 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(rs->d_qpav, q_pav, 20 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  PrepArgs a;
  a.raw = rs->d_raw;
  a.n_cols = rs->n_cols;
  a.rec_off = ts->d_rec_off;
  a.L = ts->d_L;
  a.neff_hmm = rs->d_neff_hmm;
  a.pb = rs->d_pb;
  a.R = rs->d_R;
  a.q_pav = rs->d_qpav;
  a.lg2 = c->d_lg2;
  a.diff = c->d_diff;
  a.p_tmp = rs->d_p_tmp;
  a.tr_tmp = rs->d_tr_tmp;
  a.records = ts->d_records;
  a.pav_out = rs->d_pav;
  a.gapd = par->gapd;
  a.gape = par->gape;
  a.gapf = par->gapf;
  a.gapg = par->gapg;
  a.gaph = par->gaph;
  a.gapi = par->gapi;
  a.gapb = par->gapb;
  a.pcm = par->pcm;
  a.pca = par->pca;
  a.pcb = par->pcb;
  a.columnscore = par->columnscore;
  a.ids = nullptr;
  a.lds_cols = 0;
  const int rc = launch_prepare(a, rs->d_ids, rs->n_ids, rs->max_L, c->stream);
  if (rc != 0) {
    if (!*out) hhv_tset_free(ts);
    return fail(HHV_E_DEVICE, "prepare kernel launch failed: %s", hipGetErrorString((hipError_t)(-rc)));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  rs->prepared = true;
  ts->bt_valid = false;
  ts->hits_valid = false;
  *out = ts;
  return HHV_OK;
}

int hhv_rawset_pav(hhv_ctx* c, hhv_rawset* rs, float* pav) {
  if (!c || !rs || !pav) return fail(HHV_E_ARG, "hhv_rawset_pav: null argument");
  if (!rs->prepared) return fail(HHV_E_STATE, "hhv_rawset_pav: call hhv_prepare_templates first");
  HIP_TRY(hipSetDevice(c->par.device));
  HIP_TRY(hipMemcpy(pav, rs->d_pav, (size_t)rs->n * 20 * sizeof(float), hipMemcpyDeviceToHost));
  return HHV_OK;
}

int hhv_tset_records_of(hhv_ctx* c, hhv_tset* ts, int32_t k, float* out) {
  if (!c || !ts || !out) return fail(HHV_E_ARG, "hhv_tset_records_of: null argument");
  if (k < 0 || k >= ts->n) return fail(HHV_E_ARG, "hhv_tset_records_of: template %d of %d", k, ts->n);
  HIP_TRY(hipSetDevice(c->par.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(out, ts->d_records + (size_t)ts->rec_off[k] * REC_DW, (size_t)(ts->L[k] + 1) * REC_DW * sizeof(float),
                    hipMemcpyDeviceToHost));
  return HHV_OK;
}

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
  delete db;
}

int hhv_prefilter_scores(hhv_ctx* c, hhv_pfdb* db, const uint8_t* profile, int32_t Lq, int32_t score_offset,
                         int32_t gapped, int32_t gap_init, int32_t gap_extend, const int32_t* subset, int32_t n_subset,
                         int32_t* scores) {
  if (!c || !db || !profile || !scores) return fail(HHV_E_ARG, "hhv_prefilter_scores: null argument");
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
  const size_t lds = fast ? prefilter_fast_lds(gapped != 0, Wfast) : state_lds + (generic_prof_lds ? prof_lds : 0);
  if (lds > 160 * 1024) return fail(HHV_E_LIMIT, "hhv_prefilter_scores: Lq = %d needs %zu bytes of LDS (limit 160 KiB)", Lq, lds);

  HIP_TRY(hipSetDevice(c->par.device));
  unsigned char* d_prof = nullptr;
  unsigned char* d_striped = nullptr;
  int32_t* d_subset = nullptr;
  int32_t* d_order = nullptr;
  int32_t* d_scores = nullptr;
  int rc = HHV_OK;
  std::vector<int32_t> order;
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
  if (hipMalloc(&d_prof, (size_t)220 * Lq) != hipSuccess || hipMalloc(&d_scores, (size_t)n_jobs * sizeof(int32_t)) != hipSuccess ||
      (subset && (hipMalloc(&d_subset, (size_t)n_jobs * sizeof(int32_t)) != hipSuccess ||
                  hipMalloc(&d_order, (size_t)n_jobs * sizeof(int32_t)) != hipSuccess)) ||
      (!striped.empty() && hipMalloc(&d_striped, striped.size()) != hipSuccess))
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
    a.subset = d_subset;
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
  if (rc == HHV_OK && (hipMemcpyAsync(scores, d_scores, (size_t)n_jobs * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                       hipStreamSynchronize(c->stream) != hipSuccess))
    rc = fail(HHV_E_DEVICE, "hhv_prefilter_scores: D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
  dfree(d_prof);
  dfree(d_striped);
  dfree(d_subset);
  dfree(d_order);
  dfree(d_scores);
  return rc;
}

// ---- MAC realignment (N4) --------------------------------------------------------------------------------------
struct hhv_macset {
  hhv_ctx* ctx = nullptr;
  int32_t n = 0, Lq = 0;
  std::vector<int32_t> Lt;
  std::vector<int64_t> mat_off, path_off;
  std::vector<hhv_mac_hit> hits;
  void* d_block = nullptr;  // one allocation, carved below
  size_t block_bytes = 0;
  unsigned char* d_celloff = nullptr;
  float* d_mat = nullptr;
  int32_t* d_path_i = nullptr;
  int32_t* d_path_j = nullptr;
  signed char* d_path_state = nullptr;
  float* d_path_S = nullptr;
  float* d_path_P = nullptr;
  // all five path arrays, fetched with one copy when the kernels are done
  std::vector<char> h_paths;
  size_t h_pi = 0, h_pj = 0, h_ps = 0, h_pS = 0, h_pP = 0;
};

void hhv_macset_free(hhv_macset* ms) {
  if (!ms) return;
  if (ms->ctx) (void)hipSetDevice(ms->ctx->par.device);
  if (ms->ctx && ms->d_block && ms->block_bytes > ms->ctx->mac_cache_bytes) {
    // keep the larger block for the next batch (the next round / the next query) instead of hipFree + hipMalloc
    dfree(ms->ctx->mac_cache);
    ms->ctx->mac_cache = ms->d_block;
    ms->ctx->mac_cache_bytes = ms->block_bytes;
  } else {
    dfree(ms->d_block);
  }
  delete ms;
}

struct MacMaskInput {  // what hhv_mac_realign_hits adds: masks are built on the device
  const hhv_mac_input* in = nullptr;
  int32_t n_qranges = 0, n_tranges = 0;
  const int32_t* qranges = nullptr;
  const int32_t* tranges = nullptr;
};

static int mac_realign_impl(hhv_ctx* c, const float* q_p, const float* q_tr_lin, int32_t Lq, int32_t n, const int32_t* Lt,
                            const float* const* t_p, const float* const* t_tr_lin, const uint8_t* const* celloff,
                            const MacMaskInput* mi, int32_t local, float shift, float mact, hhv_macset** out,
                            hhv_mac_hit* hits) {
  if (!c || !q_p || !q_tr_lin || !Lt || !t_p || !t_tr_lin || !out || !hits) return fail(HHV_E_ARG, "hhv_mac_realign: null argument");
  *out = nullptr;
  if (Lq < 1 || n < 1) return fail(HHV_E_ARG, "hhv_mac_realign: Lq = %d, n = %d", Lq, n);
  int max_Lt = 0;
  for (int k = 0; k < n; ++k) {
    if (Lt[k] < 1 || !t_p[k] || !t_tr_lin[k]) return fail(HHV_E_ARG, "hhv_mac_realign: bad template %d", k);
    max_Lt = std::max(max_Lt, Lt[k]);
  }
  if ((size_t)10 * (max_Lt + 2) * sizeof(double) > 160 * 1024)
    return fail(HHV_E_LIMIT, "hhv_mac_realign: template length %d exceeds the LDS row state (limit 2046)", max_Lt);
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_macset* ms = new (std::nothrow) hhv_macset();
  if (!ms) return fail(HHV_E_MEMORY, "out of host memory");
  ms->ctx = c;
  ms->n = n;
  ms->Lq = Lq;
  ms->Lt.assign(Lt, Lt + n);
  ms->mat_off.resize(n + 1);
  ms->path_off.resize(n + 1);
  std::vector<int64_t> col_off(n + 1);
  ms->mat_off[0] = ms->path_off[0] = col_off[0] = 0;
  for (int k = 0; k < n; ++k) {
    ms->mat_off[k + 1] = ms->mat_off[k] + ((int64_t)(Lq + 1) * (Lt[k] + 1) + 3) / 4 * 4;
    ms->path_off[k + 1] = ms->path_off[k] + (Lq + Lt[k] + 2 + 3) / 4 * 4;
    col_off[k + 1] = col_off[k] + Lt[k] + 1;
  }
  const int64_t cells = ms->mat_off[n], steps = ms->path_off[n], cols = col_off[n];
  // device-built masks: Viterbi paths and excluded cells, concatenated
  std::vector<int64_t> vit_off(n + 1, 0), excl_off(n + 1, 0);
  std::vector<int32_t> vit_i, vit_j, excl_i, excl_j, ends, ranges;
  if (mi) {
    for (int k = 0; k < n; ++k) {
      const hhv_mac_input& h = mi->in[k];
      if (h.nsteps < 0 || h.n_excluded < 0 || (h.nsteps > 0 && (!h.i || !h.j)) || (h.n_excluded > 0 && (!h.excluded_i || !h.excluded_j))) {
        delete ms;
        return fail(HHV_E_ARG, "hhv_mac_realign_hits: bad input %d", k);
      }
      vit_off[k + 1] = vit_off[k] + h.nsteps;
      excl_off[k + 1] = excl_off[k] + h.n_excluded;
    }
    vit_i.resize((size_t)vit_off[n] + 1);
    vit_j.resize((size_t)vit_off[n] + 1);
    excl_i.resize((size_t)excl_off[n] + 1);
    excl_j.resize((size_t)excl_off[n] + 1);
    ends.resize((size_t)n * 4);
    for (int k = 0; k < n; ++k) {
      const hhv_mac_input& h = mi->in[k];
      if (h.nsteps) {
        memcpy(&vit_i[(size_t)vit_off[k]], h.i + 1, (size_t)h.nsteps * 4);  // entries 1..nsteps
        memcpy(&vit_j[(size_t)vit_off[k]], h.j + 1, (size_t)h.nsteps * 4);
      }
      if (h.n_excluded) {
        memcpy(&excl_i[(size_t)excl_off[k]], h.excluded_i, (size_t)h.n_excluded * 4);
        memcpy(&excl_j[(size_t)excl_off[k]], h.excluded_j, (size_t)h.n_excluded * 4);
      }
      ends[(size_t)k * 4 + 0] = h.i1;
      ends[(size_t)k * 4 + 1] = h.j1;
      ends[(size_t)k * 4 + 2] = h.i2;
      ends[(size_t)k * 4 + 3] = h.j2;
    }
    for (int r = 0; r < mi->n_qranges * 2; ++r) ranges.push_back(mi->qranges[r]);
    for (int r = 0; r < mi->n_tranges * 2; ++r) ranges.push_back(mi->tranges[r]);
  }
  ranges.push_back(0);
  // carve one device allocation (256-byte aligned pieces)
  size_t total = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = total;
    total += (bytes + 255) / 256 * 256;
    return at;
  };
  const size_t o_qp = carve((size_t)(Lq + 1) * 20 * 4), o_qtr = carve((size_t)(Lq + 1) * 7 * 4), o_tp = carve((size_t)cols * 20 * 4),
               o_ttr = carve((size_t)cols * 7 * 4), o_col = carve((size_t)n * 8), o_Lt = carve((size_t)n * 4),
               o_moff = carve((size_t)n * 8), o_co = carve((size_t)cells), o_mat = carve((size_t)cells * 4),
               o_bmm = carve((size_t)cells), o_scale = carve((size_t)n * (Lq + 2) * 8), o_pf = carve((size_t)n * 8),
               o_hits = carve((size_t)n * sizeof(DevMacHit)), o_poff = carve((size_t)n * 8), o_pi = carve((size_t)steps * 4),
               o_pj = carve((size_t)steps * 4), o_ps = carve((size_t)steps), o_pS = carve((size_t)steps * 4),
               o_pP = carve((size_t)steps * 4);
  const size_t path_bytes = total - o_pi;
  const size_t o_ends = carve(ends.size() * 4 + 16), o_voff = carve((size_t)(n + 1) * 8), o_vi = carve(vit_i.size() * 4 + 4),
               o_vj = carve(vit_j.size() * 4 + 4), o_xoff = carve((size_t)(n + 1) * 8), o_xi = carve(excl_i.size() * 4 + 4),
               o_xj = carve(excl_j.size() * 4 + 4), o_rg = carve(ranges.size() * 4);
  if (c->mac_cache && c->mac_cache_bytes >= total) {
    ms->d_block = c->mac_cache;
    ms->block_bytes = c->mac_cache_bytes;
    c->mac_cache = nullptr;
    c->mac_cache_bytes = 0;
  } else if (hipMalloc(&ms->d_block, total) != hipSuccess) {
    delete ms;
    return fail(HHV_E_MEMORY, "hhv_mac_realign: cannot allocate %zu bytes on the device", total);
  } else {
    ms->block_bytes = total;
  }
  char* base = (char*)ms->d_block;
  // host staging of the ragged inputs
  std::vector<float> tp((size_t)cols * 20), ttr((size_t)cols * 7);
  std::vector<unsigned char> co;
  if (!mi) co.assign((size_t)cells, 0);
  for (int k = 0; k < n; ++k) {
    memcpy(&tp[(size_t)col_off[k] * 20], t_p[k], (size_t)(Lt[k] + 1) * 20 * 4);
    memcpy(&ttr[(size_t)col_off[k] * 7], t_tr_lin[k], (size_t)(Lt[k] + 1) * 7 * 4);
    if (!mi && celloff && celloff[k]) {
      unsigned char* dst = &co[(size_t)ms->mat_off[k]];
      const uint8_t* src = celloff[k];
      for (size_t e = 0; e < (size_t)(Lq + 1) * (Lt[k] + 1); ++e) dst[e] = src[e] ? 1 : 0;
    }
  }
  int rc = HHV_OK;
  hipStream_t st = c->stream;
  if (hipMemcpyAsync(base + o_qp, q_p, (size_t)(Lq + 1) * 20 * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(base + o_qtr, q_tr_lin, (size_t)(Lq + 1) * 7 * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(base + o_tp, tp.data(), tp.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(base + o_ttr, ttr.data(), ttr.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(base + o_col, col_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(base + o_Lt, Lt, (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(base + o_moff, ms->mat_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
      (!mi && hipMemcpyAsync(base + o_co, co.data(), co.size(), hipMemcpyHostToDevice, st) != hipSuccess) ||
      (mi && (hipMemcpyAsync(base + o_ends, ends.data(), ends.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
              hipMemcpyAsync(base + o_voff, vit_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
              hipMemcpyAsync(base + o_vi, vit_i.data(), vit_i.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
              hipMemcpyAsync(base + o_vj, vit_j.data(), vit_j.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
              hipMemcpyAsync(base + o_xoff, excl_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
              hipMemcpyAsync(base + o_xi, excl_i.data(), excl_i.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
              hipMemcpyAsync(base + o_xj, excl_j.data(), excl_j.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
              hipMemcpyAsync(base + o_rg, ranges.data(), ranges.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess)) ||
      hipMemcpyAsync(base + o_poff, ms->path_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess)
    rc = fail(HHV_E_DEVICE, "hhv_mac_realign: H2D copy failed");
  MacArgs a;
  a.n = n;
  a.Lq = Lq;
  a.q_p = (const float*)(base + o_qp);
  a.q_tr = (const float*)(base + o_qtr);
  a.t_p = (const float*)(base + o_tp);
  a.t_tr = (const float*)(base + o_ttr);
  a.col_off = (const int64_t*)(base + o_col);
  a.Lt = (const int32_t*)(base + o_Lt);
  a.mat_off = (const int64_t*)(base + o_moff);
  a.celloff = (const unsigned char*)(base + o_co);
  a.mat = (float*)(base + o_mat);
  a.bmm = (unsigned char*)(base + o_bmm);
  a.scale = (double*)(base + o_scale);
  a.Pforward = (double*)(base + o_pf);
  a.hits = (DevMacHit*)(base + o_hits);
  a.Cshift = pow(2.0, (double)shift);  // src/hhforwardalgorithm.cpp:15
  a.mact = mact;
  a.path_off = (const int64_t*)(base + o_poff);
  a.path_i = (int32_t*)(base + o_pi);
  a.path_j = (int32_t*)(base + o_pj);
  a.path_state = (signed char*)(base + o_ps);
  a.path_S = (float*)(base + o_pS);
  a.path_P = (float*)(base + o_pP);
  a.lg2 = c->d_lg2;
  a.diff = c->d_diff;
  ms->d_mat = a.mat;
  ms->d_celloff = (unsigned char*)(base + o_co);
  ms->d_path_i = a.path_i;
  ms->d_path_j = a.path_j;
  ms->d_path_state = a.path_state;
  ms->d_path_S = a.path_S;
  ms->d_path_P = a.path_P;
  if (rc == HHV_OK) {
    (void)hipEventRecord(c->ev0, st);
    int lr = 0;
    if (mi) {
      MacMaskArgs m;
      m.ends = (const int4*)(base + o_ends);
      m.vit_off = (const int64_t*)(base + o_voff);
      m.vit_i = (const int32_t*)(base + o_vi);
      m.vit_j = (const int32_t*)(base + o_vj);
      m.excl_off = (const int64_t*)(base + o_xoff);
      m.excl_i = (const int32_t*)(base + o_xi);
      m.excl_j = (const int32_t*)(base + o_xj);
      m.ranges = (const int32_t*)(base + o_rg);
      m.n_qranges = mi->n_qranges;
      m.n_tranges = mi->n_tranges;
      lr = launch_mac_mask(a, m, st);
    }
    if (lr == 0) lr = launch_mac(a, local != 0, max_Lt, st);
    (void)hipEventRecord(c->ev1, st);
    c->ev_valid = true;
    if (lr != 0) rc = fail(HHV_E_DEVICE, "MAC kernel launch failed: %s", hipGetErrorString((hipError_t)(-lr)));
  }
  static_assert(sizeof(DevMacHit) == sizeof(hhv_mac_hit), "hhv_mac_hit layout");
  ms->hits.resize(n);
  ms->h_paths.resize(path_bytes);
  ms->h_pi = 0;
  ms->h_pj = o_pj - o_pi;
  ms->h_ps = o_ps - o_pi;
  ms->h_pS = o_pS - o_pi;
  ms->h_pP = o_pP - o_pi;
  if (rc == HHV_OK && (hipMemcpyAsync(ms->hits.data(), base + o_hits, (size_t)n * sizeof(hhv_mac_hit), hipMemcpyDeviceToHost, st) != hipSuccess ||
                       hipMemcpyAsync(ms->h_paths.data(), base + o_pi, path_bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
                       hipStreamSynchronize(st) != hipSuccess))
    rc = fail(HHV_E_DEVICE, "hhv_mac_realign: kernels failed: %s", hipGetErrorString(hipGetLastError()));
  if (rc != HHV_OK) {
    hhv_macset_free(ms);
    return rc;
  }
  memcpy(hits, ms->hits.data(), (size_t)n * sizeof(hhv_mac_hit));
  *out = ms;
  return HHV_OK;
}

int hhv_mac_realign(hhv_ctx* c, const float* q_p, const float* q_tr_lin, int32_t Lq, int32_t n, const int32_t* Lt,
                    const float* const* t_p, const float* const* t_tr_lin, const uint8_t* const* celloff, int32_t local,
                    float shift, float mact, hhv_macset** out, hhv_mac_hit* hits) {
  return mac_realign_impl(c, q_p, q_tr_lin, Lq, n, Lt, t_p, t_tr_lin, celloff, nullptr, local, shift, mact, out, hits);
}

int hhv_mac_realign_hits(hhv_ctx* c, const float* q_p, const float* q_tr_lin, int32_t Lq, int32_t n, const int32_t* Lt,
                         const float* const* t_p, const float* const* t_tr_lin, const hhv_mac_input* in, int32_t n_qranges,
                         const int32_t* qranges, int32_t n_tranges, const int32_t* tranges, int32_t local, float shift,
                         float mact, hhv_macset** out, hhv_mac_hit* hits) {
  if (!in || n_qranges < 0 || n_tranges < 0 || (n_qranges && !qranges) || (n_tranges && !tranges))
    return fail(HHV_E_ARG, "hhv_mac_realign_hits: bad argument");
  MacMaskInput mi;
  mi.in = in;
  mi.n_qranges = n_qranges;
  mi.qranges = qranges;
  mi.n_tranges = n_tranges;
  mi.tranges = tranges;
  return mac_realign_impl(c, q_p, q_tr_lin, Lq, n, Lt, t_p, t_tr_lin, nullptr, &mi, local, shift, mact, out, hits);
}

int hhv_mac_celloff(hhv_macset* ms, int32_t k, uint8_t* mask) {
  if (!ms || k < 0 || k >= ms->n || !mask) return fail(HHV_E_ARG, "hhv_mac_celloff: bad argument");
  HIP_TRY(hipSetDevice(ms->ctx->par.device));
  HIP_TRY(hipMemcpy(mask, ms->d_celloff + ms->mat_off[k], (size_t)(ms->Lq + 1) * (ms->Lt[k] + 1), hipMemcpyDeviceToHost));
  return HHV_OK;
}

int hhv_mac_path(hhv_macset* ms, int32_t k, int32_t cap, int32_t* i_steps, int32_t* j_steps, int8_t* states, float* S,
                 float* P_posterior, int32_t* nsteps) {
  if (!ms || k < 0 || k >= ms->n || !nsteps) return fail(HHV_E_ARG, "hhv_mac_path: bad argument");
  const int ns = ms->hits[k].nsteps;
  *nsteps = ns;
  if (cap < ns + 1) return fail(HHV_E_ARG, "hhv_mac_path: cap %d < nsteps + 1 = %d", cap, ns + 1);
  const int64_t o = ms->path_off[k];
  const size_t cnt = (size_t)ns + 1;
  const char* hp = ms->h_paths.data();
  if (i_steps) memcpy(i_steps, hp + ms->h_pi + (size_t)o * 4, cnt * 4);
  if (j_steps) memcpy(j_steps, hp + ms->h_pj + (size_t)o * 4, cnt * 4);
  if (states) memcpy(states, hp + ms->h_ps + (size_t)o, cnt);
  if (S) memcpy(S, hp + ms->h_pS + (size_t)o * 4, cnt * 4);
  if (P_posterior) memcpy(P_posterior, hp + ms->h_pP + (size_t)o * 4, cnt * 4);
  return HHV_OK;
}

int hhv_mac_posterior(hhv_macset* ms, int32_t k, float* posterior) {
  if (!ms || k < 0 || k >= ms->n || !posterior) return fail(HHV_E_ARG, "hhv_mac_posterior: bad argument");
  HIP_TRY(hipSetDevice(ms->ctx->par.device));
  HIP_TRY(hipMemcpy(posterior, ms->d_mat + ms->mat_off[k], (size_t)(ms->Lq + 1) * (ms->Lt[k] + 1) * 4, hipMemcpyDeviceToHost));
  return HHV_OK;
}

// ---- binary packed template database (N1) -------------------------------------------------------
namespace {
struct DbHeader {
  char magic[8];
  int32_t n;
  int32_t record_dwords;
  int64_t n_records;
  char pad[40];
};
static_assert(sizeof(DbHeader) == 64, "db header");
}  // namespace

int hhv_db_write(const char* path, int32_t n, const int32_t* L, const float* const* p, const float* const* tr,
                 const int8_t* const* ss_pred, const int8_t* const* ss_conf, const int8_t* const* ss_dssp) {
  if (!path || !L || !p || !tr || n < 1) return fail(HHV_E_ARG, "hhv_db_write: bad argument");
  int64_t nrec = 1;
  for (int k = 0; k < n; ++k) {
    if (L[k] < 1 || L[k] > 0xFFFF || !p[k] || !tr[k]) return fail(HHV_E_ARG, "hhv_db_write: template %d invalid", k);
    nrec += (int64_t)L[k] + 1;
  }
  FILE* f = fopen(path, "wb");
  if (!f) return fail(HHV_E_ARG, "hhv_db_write: cannot open %s", path);
  DbHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "HHVPDB01", 8);
  h.n = n;
  h.record_dwords = REC_DW;
  h.n_records = nrec;
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(L, sizeof(int32_t), (size_t)n, f) == (size_t)n;
  std::vector<float> buf;
  for (int k = 0; k < n && ok; ++k) {
    buf.resize(((size_t)L[k] + 1) * REC_DW);
    pack_template(p[k], tr[k], L[k], k, buf.data(), ss_pred ? ss_pred[k] : nullptr, ss_conf ? ss_conf[k] : nullptr,
                  ss_dssp ? ss_dssp[k] : nullptr);
    ok = fwrite(buf.data(), sizeof(float), buf.size(), f) == buf.size();
  }
  if (ok) {
    buf.assign(REC_DW, 0.0f);
    write_header(buf.data(), -1, 0);
    ok = fwrite(buf.data(), sizeof(float), REC_DW, f) == (size_t)REC_DW;
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? HHV_OK : fail(HHV_E_ARG, "hhv_db_write: write to %s failed", path);
}

int hhv_db_open(hhv_ctx* c, const char* path, hhv_tset** out) {
  if (!c || !path || !out) return fail(HHV_E_ARG, "hhv_db_open: null argument");
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return fail(HHV_E_ARG, "hhv_db_open: cannot open %s", path);
  DbHeader h;
  if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "HHVPDB01", 8) != 0 || h.record_dwords != REC_DW || h.n < 1) {
    fclose(f);
    return fail(HHV_E_ARG, "hhv_db_open: %s is not a packed template database", path);
  }
  std::vector<int32_t> L((size_t)h.n);
  if (fread(L.data(), sizeof(int32_t), L.size(), f) != L.size()) {
    fclose(f);
    return fail(HHV_E_ARG, "hhv_db_open: truncated length table");
  }
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_tset* ts = new (std::nothrow) hhv_tset();
  if (!ts) {
    fclose(f);
    return fail(HHV_E_MEMORY, "out of host memory");
  }
  int rc = tset_init_common(c, ts, h.n, L.data());
  if (rc == HHV_OK && ts->n_records != h.n_records) rc = fail(HHV_E_ARG, "hhv_db_open: record count mismatch");
  if (rc == HHV_OK && hipMalloc(&ts->d_records, (size_t)(ts->n_records + STREAM_PAD_RECS) * REC_DW * sizeof(float)) != hipSuccess)
    rc = fail(HHV_E_MEMORY, "hhv_db_open: device allocation failed");
  if (rc == HHV_OK) {
    ts->owns_records = true;
    (void)hipMemset(ts->d_records + (size_t)ts->n_records * REC_DW, 0, (size_t)STREAM_PAD_RECS * REC_DW * sizeof(float));
    const size_t slab = (64u << 20) / sizeof(float);
    std::vector<float> buf(slab);
    size_t left = (size_t)ts->n_records * REC_DW, off = 0;
    while (left && rc == HHV_OK) {
      const size_t m = std::min(left, slab);
      if (fread(buf.data(), sizeof(float), m, f) != m) rc = fail(HHV_E_ARG, "hhv_db_open: truncated record stream");
      else if (hipMemcpy(ts->d_records + off, buf.data(), m * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(HHV_E_DEVICE, "hhv_db_open: H2D copy failed");
      off += m;
      left -= m;
    }
  }
  fclose(f);
  if (rc != HHV_OK) {
    hhv_tset_free(ts);
    return rc;
  }
  *out = ts;
  return HHV_OK;
}

void hhv_tset_free(hhv_tset* ts) {
  if (!ts) return;
  if (ts->ctx) (void)hipSetDevice(ts->ctx->par.device);
  if (ts->owns_records) dfree(ts->d_records);
  dfree(ts->d_rec_off);
  dfree(ts->d_L);
  dfree(ts->d_results);
  dfree(ts->d_wave_rec);
  dfree(ts->d_bt);
  dfree(ts->d_carry);
  dfree(ts->d_carry_mi);
  dfree(ts->d_path_off);
  dfree(ts->d_i_steps);
  dfree(ts->d_j_steps);
  dfree(ts->d_states);
  dfree(ts->d_S);
  dfree(ts->d_hits);
  dfree(ts->d_topk);
  dfree(ts->d_keys);
  dfree(ts->d_sorted);
  dfree(ts->d_sort_temp);
  dfree(ts->d_raw_hits);
  delete ts;
}

int32_t hhv_tset_size(const hhv_tset* ts) { return ts ? ts->n : 0; }
int64_t hhv_tset_records(const hhv_tset* ts) { return ts ? ts->n_records : 0; }
int64_t hhv_tset_cells(const hhv_tset* ts, int32_t Lq) {
  if (!ts) return 0;
  int64_t s = 0;
  for (int32_t l : ts->L) s += (int64_t)Lq * l;
  return s;
}

// Contiguous template ranges with ~equal record counts, one per wave.  All waves are resident at
// once (n_waves = CUs x blocks/CU the variant's VGPR/LDS budget admits), so there is no tail.
static int ensure_partition(hhv_ctx* c, hhv_tset* ts, int n_waves) {
  if (ts->n_waves == n_waves && ts->d_wave_rec) return HHV_OK;
  dfree(ts->d_wave_rec);
  std::vector<int64_t> wr((size_t)n_waves + 1);
  const int64_t total = ts->rec_off[ts->n];
  int k = 0;
  for (int w = 0; w < n_waves; ++w) {
    const int64_t target = (int64_t)(((__int128)total * w) / n_waves);
    while (k < ts->n && ts->rec_off[k] < target) ++k;
    wr[w] = ts->rec_off[k];
  }
  wr[n_waves] = total;
  HIP_TRY(hipMalloc(&ts->d_wave_rec, wr.size() * sizeof(int64_t)));
  HIP_TRY(hipMemcpyAsync(ts->d_wave_rec, wr.data(), wr.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  ts->n_waves = n_waves;
  return HHV_OK;
}

static int ensure_bt(hhv_ctx* c, hhv_tset* ts) {
  if (ts->d_bt && ts->bt_R == c->R && ts->bt_P == c->P) return HHV_OK;
  dfree(ts->d_bt);
  const size_t bytes = (size_t)c->P * ts->n_records * LANES * sizeof(uint64_t);
  if (hipMalloc(&ts->d_bt, bytes) != hipSuccess)
    return fail(HHV_E_MEMORY, "backtrace buffer of %zu bytes does not fit on the device", bytes);
  HIP_TRY(hipMemsetAsync(ts->d_bt, 0, bytes, c->stream));
  ts->bt_R = c->R;
  ts->bt_P = c->P;
  ts->bt_valid = false;
  return HHV_OK;
}

int hhv_align_async(hhv_ctx* c, hhv_tset* ts, uint32_t flags, void* d_out) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_align: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_align: template set belongs to another context");
  if (c->Lq < 1) return fail(HHV_E_STATE, "hhv_align: no query set");
  HIP_TRY(hipSetDevice(c->par.device));
  const bool celloff = (flags & HHV_ALIGN_CELLOFF) != 0;
  const bool bt = celloff || (flags & HHV_ALIGN_BACKTRACE) != 0;
  const bool local = c->par.local != 0;
  int rc = ensure_ss(c);
  if (rc != HHV_OK) return rc;
  // src/hhviterbi.cpp:175: the ...AndSS kernels run only for ssm == SCORE_ALIGNMENT and a non-zero ss_hmm_mode
  const bool ss = c->par.ss_mode == 2 && c->ss_hmm_mode != 0;
  int blocks_per_cu = 0, vgprs = 0;
  rc = stream_kernel_occupancy(c->R, local, bt, celloff, c->P > 1, ss, &blocks_per_cu, &vgprs);
  if (rc != 0 || blocks_per_cu < 1) return fail(HHV_E_DEVICE, "occupancy query failed (%d)", rc);
  const int n_waves = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)c->num_cus * blocks_per_cu, ts->n));
  rc = ensure_partition(c, ts, n_waves);
  if (rc != HHV_OK) return rc;
  if (bt) {
    rc = ensure_bt(c, ts);
    if (rc != HHV_OK) return rc;
  }
  StreamArgs a;
  a.records = ts->d_records;
  a.wave_rec = ts->d_wave_rec;
  a.qpack = c->d_qpack;
  a.results = d_out ? (DevResult*)d_out : ts->d_results;
  a.bt = ts->d_bt;
  a.egq = c->par.egq;
  a.egt = c->par.egt;
  a.shift = c->par.shift;
  a.Lq = c->Lq;
  a.carry = nullptr;
  a.carry_mi = nullptr;
  a.bt_pass_stride = ts->n_records * LANES;
  a.ss_table = ss ? c->d_ss_table : nullptr;
  a.ss_q_off = ss ? c->d_ss_q_off : nullptr;
  a.ss_t_shift = c->ss_t_shift;
  a.ss_t_mask = c->ss_t_mask;
  if (c->P > 1) {
    if (!ts->d_carry) {
      HIP_TRY(hipMalloc(&ts->d_carry, (size_t)ts->n_records * sizeof(float4)));
      HIP_TRY(hipMalloc(&ts->d_carry_mi, (size_t)ts->n_records * sizeof(float)));
    }
    a.carry = ts->d_carry;
    a.carry_mi = ts->d_carry_mi;
  }
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  for (int pass = 0; pass < c->P; ++pass) {
    a.qpack = c->d_qpack + (size_t)pass * LANES * c->R * REC_DW;
    a.row_base = pass * LANES * c->R;
    a.pass_first = pass == 0;
    a.pass_last = pass == c->P - 1;
    rc = launch_stream(c->R, local, bt, celloff, c->P > 1, ss, a, n_waves, c->stream);
    if (rc != 0) return fail(HHV_E_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)(-rc)));
  }
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  c->ev_valid = true;
  if (d_out) {
    HIP_TRY(hipMemcpyAsync(ts->d_results, d_out, (size_t)ts->n * sizeof(DevResult), hipMemcpyDeviceToDevice, c->stream));
  }
  if (bt) {
    ts->bt_valid = true;
    ts->bt_Lq = c->Lq;
  }
  ts->hits_valid = false;
  return HHV_OK;
}

int hhv_sync(hhv_ctx* c) {
  if (!c) return fail(HHV_E_ARG, "hhv_sync: null");
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HHV_OK;
}

void* hhv_stream(hhv_ctx* c) { return c ? (void*)c->stream : nullptr; }

int hhv_last_kernel_ms(hhv_ctx* c, float* ms) {
  if (!c || !ms) return fail(HHV_E_ARG, "hhv_last_kernel_ms: null");
  if (!c->ev_valid) return fail(HHV_E_STATE, "hhv_last_kernel_ms: no launch yet");
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return HHV_OK;
}

int hhv_align(hhv_ctx* c, hhv_tset* ts, uint32_t flags, hhv_result* out) {
  int rc = hhv_align_async(c, ts, flags, nullptr);
  if (rc != HHV_OK) return rc;
  if (out) {
    HIP_TRY(hipMemcpyAsync(out, ts->d_results, (size_t)ts->n * sizeof(DevResult), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HHV_OK;
}

int hhv_set_celloff(hhv_ctx* c, hhv_tset* ts, int32_t k, const uint8_t* mask) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_set_celloff: null argument");
  if (k < 0 || k >= ts->n) return fail(HHV_E_ARG, "hhv_set_celloff: template %d of %d", k, ts->n);
  if (c->Lq < 1) return fail(HHV_E_STATE, "hhv_set_celloff: no query set");
  HIP_TRY(hipSetDevice(c->par.device));
  int rc = ensure_bt(c, ts);
  if (rc != HHV_OK) return rc;
  const int Lt = ts->L[k], Lq = c->Lq, R = c->R;
  // per pass: entries of columns 1..Lt: [Lt][64] x 8 bytes; only bit 7 of each byte is an input of the kernel
  std::vector<uint64_t> e((size_t)Lt * LANES);
  for (int pass = 0; pass < c->P; ++pass) {
    std::fill(e.begin(), e.end(), 0);
    if (mask) {
      const int ilo = pass * LANES * R + 1, ihi = std::min(Lq, (pass + 1) * LANES * R);
      for (int i = ilo; i <= ihi; ++i) {
        const int g = (i - ilo) / R, r = (i - ilo) % R;
        const uint8_t* row = mask + (size_t)i * (Lt + 1);
        for (int j = 1; j <= Lt; ++j)
          if (row[j]) e[(size_t)(j - 1) * LANES + g] |= (uint64_t)0x80 << (8 * r);
      }
    }
    HIP_TRY(hipMemcpyAsync(ts->d_bt + (size_t)pass * ts->n_records * LANES + (size_t)(ts->rec_off[k] + 1) * LANES,
                           e.data(), e.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  ts->bt_valid = false;
  return HHV_OK;
}

int hhv_backtrace_matrix(hhv_ctx* c, hhv_tset* ts, int32_t k, uint8_t* out) {
  if (!c || !ts || !out) return fail(HHV_E_ARG, "hhv_backtrace_matrix: null argument");
  if (k < 0 || k >= ts->n) return fail(HHV_E_ARG, "hhv_backtrace_matrix: template %d of %d", k, ts->n);
  if (!ts->bt_valid || !ts->d_bt) return fail(HHV_E_STATE, "hhv_backtrace_matrix: no backtrace computed");
  HIP_TRY(hipSetDevice(c->par.device));
  const int Lt = ts->L[k], Lq = ts->bt_Lq, R = ts->bt_R;
  std::vector<uint64_t> e((size_t)Lt * LANES);
  HIP_TRY(hipStreamSynchronize(c->stream));
  memset(out, 0, (size_t)(Lq + 1) * (Lt + 1));
  for (int pass = 0; pass < ts->bt_P; ++pass) {
    HIP_TRY(hipMemcpy(e.data(), ts->d_bt + (size_t)pass * ts->n_records * LANES + (size_t)(ts->rec_off[k] + 1) * LANES,
                      e.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    const int ilo = pass * LANES * R + 1, ihi = std::min(Lq, (pass + 1) * LANES * R);
    for (int i = ilo; i <= ihi; ++i) {
      const int g = (i - ilo) / R, r = (i - ilo) % R;
      uint8_t* row = out + (size_t)i * (Lt + 1);
      for (int j = 1; j <= Lt; ++j) row[j] = (uint8_t)bt_decode(e[(size_t)(j - 1) * LANES + g], r, R);
    }
  }
  return HHV_OK;
}

static int ensure_paths(hhv_ctx* c, hhv_tset* ts) {
  if (ts->path_Lq == c->Lq && ts->d_hits) return HHV_OK;
  dfree(ts->d_path_off);
  dfree(ts->d_i_steps);
  dfree(ts->d_j_steps);
  dfree(ts->d_states);
  dfree(ts->d_S);
  dfree(ts->d_hits);
  ts->path_off.resize((size_t)ts->n + 1);
  int64_t off = 0;
  for (int k = 0; k < ts->n; ++k) {
    ts->path_off[k] = off;
    off += (int64_t)c->Lq + ts->L[k] + 2;  // BacktraceResult arrays: i2 + j2 + 2 entries (src/hhviterbi.cpp:90-93)
  }
  ts->path_off[ts->n] = off;
  HIP_TRY(hipMalloc(&ts->d_path_off, ts->path_off.size() * sizeof(int64_t)));
  HIP_TRY(hipMalloc(&ts->d_i_steps, (size_t)off * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&ts->d_j_steps, (size_t)off * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&ts->d_states, (size_t)off * sizeof(int8_t)));
  HIP_TRY(hipMalloc(&ts->d_S, (size_t)off * sizeof(float)));
  HIP_TRY(hipMalloc(&ts->d_hits, (size_t)ts->n * sizeof(DevHit)));
  HIP_TRY(hipMemcpy(ts->d_path_off, ts->path_off.data(), ts->path_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  ts->path_Lq = c->Lq;
  return HHV_OK;
}

static int run_trace(hhv_ctx* c, hhv_tset* ts) {
  if (!ts->bt_valid || !ts->d_bt || ts->bt_Lq != c->Lq)
    return fail(HHV_E_STATE, "hhv_hits: needs a preceding hhv_align with HHV_ALIGN_BACKTRACE for the current query");
  int rc = ensure_paths(c, ts);
  if (rc != HHV_OK) return rc;
  TraceArgs a;
  a.records = ts->d_records;
  a.rec_off = ts->d_rec_off;
  a.L = ts->d_L;
  a.qp = c->d_qp;
  a.results = ts->d_results;
  a.bt = ts->d_bt;
  a.lg2 = c->d_lg2;
  a.diff = c->d_diff;
  a.hits = ts->d_hits;
  a.i_steps = ts->d_i_steps;
  a.j_steps = ts->d_j_steps;
  a.states = ts->d_states;
  a.S = ts->d_S;
  a.path_off = ts->d_path_off;
  a.corr = c->par.corr;
  a.ss_mode = c->par.ss_mode;
  a.Lq = c->Lq;
  a.R = ts->bt_R;
  a.n = ts->n;
  a.bt_pass_stride = ts->n_records * LANES;
  rc = ensure_ss(c);
  if (rc != HHV_OK) return rc;
  a.ss_table = c->ss_hmm_mode ? c->d_ss_table : nullptr;
  a.ss_q_off = c->ss_hmm_mode ? c->d_ss_q_off : nullptr;
  a.ss_t_shift = c->ss_t_shift;
  a.ss_t_mask = c->ss_t_mask;
  rc = launch_trace(a, c->stream);
  if (rc != 0) return fail(HHV_E_DEVICE, "trace kernel launch failed: %s", hipGetErrorString((hipError_t)(-rc)));
  ts->hits_valid = true;
  return HHV_OK;
}

int hhv_hits(hhv_ctx* c, hhv_tset* ts, hhv_hit* hits) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_hits: null argument");
  HIP_TRY(hipSetDevice(c->par.device));
  int rc = run_trace(c, ts);
  if (rc != HHV_OK) return rc;
  if (hits) HIP_TRY(hipMemcpyAsync(hits, ts->d_hits, (size_t)ts->n * sizeof(DevHit), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HHV_OK;
}

int hhv_hit_path(hhv_ctx* c, hhv_tset* ts, int32_t k, int32_t cap, int32_t* i_steps, int32_t* j_steps, int8_t* states,
                 float* S, int32_t* nsteps) {
  if (!c || !ts || !nsteps) return fail(HHV_E_ARG, "hhv_hit_path: null argument");
  if (k < 0 || k >= ts->n) return fail(HHV_E_ARG, "hhv_hit_path: template %d of %d", k, ts->n);
  if (!ts->hits_valid) return fail(HHV_E_STATE, "hhv_hit_path: call hhv_hits first");
  HIP_TRY(hipSetDevice(c->par.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  DevHit h;
  HIP_TRY(hipMemcpy(&h, ts->d_hits + k, sizeof(h), hipMemcpyDeviceToHost));
  *nsteps = h.nsteps;
  const int m = std::min(cap, h.nsteps + 1);
  if (m <= 0) return HHV_OK;
  const int64_t po = ts->path_off[k];
  if (i_steps) HIP_TRY(hipMemcpy(i_steps, ts->d_i_steps + po, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (j_steps) HIP_TRY(hipMemcpy(j_steps, ts->d_j_steps + po, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (states) HIP_TRY(hipMemcpy(states, ts->d_states + po, (size_t)m * sizeof(int8_t), hipMemcpyDeviceToHost));
  if (S) HIP_TRY(hipMemcpy(S, ts->d_S + po, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
  if (i_steps) i_steps[0] = 0;
  if (j_steps) j_steps[0] = 0;
  if (states) states[0] = 0;
  if (S) S[0] = 0.0f;
  return HHV_OK;
}

int hhv_topk(hhv_ctx* c, hhv_tset* ts, int32_t k, uint32_t flags, hhv_hit* out, void* d_out, int32_t* n_out) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_topk: null argument");
  if (k < 1) return fail(HHV_E_ARG, "hhv_topk: k = %d", k);
  const bool raw = (flags & HHV_TOPK_RAW) != 0;
  if (!raw && !ts->hits_valid) return fail(HHV_E_STATE, "hhv_topk: call hhv_hits first (or pass HHV_TOPK_RAW)");
  HIP_TRY(hipSetDevice(c->par.device));
  const int kk = std::min(k, ts->n);
  if (ts->topk_cap < k) {
    dfree(ts->d_topk);
    HIP_TRY(hipMalloc(&ts->d_topk, (size_t)k * sizeof(DevHit)));
    ts->topk_cap = k;
  }
  if (!ts->d_keys) {
    HIP_TRY(hipMalloc(&ts->d_keys, (size_t)ts->n * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&ts->d_sorted, (size_t)ts->n * sizeof(uint64_t)));
    ts->sort_temp_bytes = topk_temp_bytes(ts->n);
    HIP_TRY(hipMalloc(&ts->d_sort_temp, ts->sort_temp_bytes));
  }
  const DevHit* src = ts->d_hits;
  if (raw) {
    if (!ts->d_raw_hits) HIP_TRY(hipMalloc(&ts->d_raw_hits, (size_t)ts->n * sizeof(DevHit)));
    results_to_hits(ts->d_results, ts->n, ts->d_raw_hits, c->stream);
    src = ts->d_raw_hits;
  }
  DevHit* dst = d_out ? (DevHit*)d_out : ts->d_topk;
  std::string err;
  if (topk_device(src, ts->n, kk, dst, ts->d_keys, ts->d_sorted, ts->d_sort_temp, ts->sort_temp_bytes, c->stream,
                  &err) != 0)
    return fail(HHV_E_DEVICE, "hhv_topk: %s", err.c_str());
  if (kk < k) HIP_TRY(hipMemsetAsync(dst + kk, 0xFF, (size_t)(k - kk) * sizeof(DevHit), c->stream));
  if (out) HIP_TRY(hipMemcpyAsync(out, dst, (size_t)k * sizeof(DevHit), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (n_out) *n_out = kk;
  return HHV_OK;
}

}  // extern "C"
