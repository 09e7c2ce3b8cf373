// hhv_api_mac.cpp -- C ABI of the MAC realignment.
#include "hhv_api_common.h"

#include <algorithm>
#include <cmath>

using namespace hhv;
using hhv::api::dfree;
using hhv::api::fail;
using hhv::api::tset_init_common;

extern "C" {

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
  float* d_fwd_list = nullptr;  // hhv_mac_set_lists: dense planes of the forward / backward list values (else null)
  float* d_bwd_list = nullptr;
  int32_t* d_path_i = nullptr;
  int32_t* d_path_j = nullptr;
  signed char* d_path_state = nullptr;
  float* d_path_S = nullptr;
  float* d_path_P = nullptr;
  // all five path arrays, fetched with one copy when the kernels are done, into a PINNED buffer that is handed back to the
  // context when the set is freed (a pageable 5 MB vector cost a zero fill plus a staged copy: ~1.5 ms per 500 hits)
  char* h_paths = nullptr;
  size_t h_paths_bytes = 0;
  size_t h_pi = 0, h_pj = 0, h_ps = 0, h_pS = 0, h_pP = 0;
  // hhv_mac_list: the list built last (hit, which) - a caller asks twice, for the count and for the entries (each
  // call used to copy the dense plane of the hit again)
  int32_t list_k = -1, list_which = -1;
  std::vector<int32_t> list_i, list_j;
  std::vector<float> list_v;
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
  if (ms->h_paths) {
    if (ms->ctx && ms->h_paths_bytes > ms->ctx->mac_pinned_out_bytes) {
      if (ms->ctx->mac_pinned_out) (void)hipHostFree(ms->ctx->mac_pinned_out);
      ms->ctx->mac_pinned_out = ms->h_paths;
      ms->ctx->mac_pinned_out_bytes = ms->h_paths_bytes;
    } else {
      (void)hipHostFree(ms->h_paths);
    }
  }
  delete ms;
}

struct MacMaskInput {  // what hhv_mac_realign_hits adds: masks are built on the device
  const hhv_tset* ts = nullptr;         // hhv_mac_realign_tset: profiles are read from this resident set ...
  const int32_t* template_of = nullptr; // ... hit k is template template_of[k] of it
  const hhv_mac_input* in = nullptr;
  int32_t n_qranges = 0, n_tranges = 0;
  const int32_t* qranges = nullptr;
  const int32_t* tranges = nullptr;
};

static int mac_realign_impl(hhv_ctx* c, const float* q_p, const float* q_tr_lin, int32_t Lq, int32_t n, const int32_t* Lt,
                            const float* const* t_p, const float* const* t_tr_lin, const uint8_t* const* celloff,
                            const MacMaskInput* mi, int32_t local, float shift, float mact, hhv_macset** out,
                            hhv_mac_hit* hits) {
  const bool from_tset = mi && mi->ts;
  if (!c || !q_p || !q_tr_lin || !Lt || (!from_tset && !t_p) || !t_tr_lin || !out || !hits)
    return fail(HHV_E_ARG, "hhv_mac_realign: null argument");
  *out = nullptr;
  if (Lq < 1 || n < 1) return fail(HHV_E_ARG, "hhv_mac_realign: Lq = %d, n = %d", Lq, n);
  MacClasses cls = {};  // the launch is split by template length (hhv_internal.h)
  // first by length alone; then the longest hits keep their dataflow classes for as long as the GPU holds them at once
  // (mac_dataflow_budget), the others go to the class without LDS
  std::vector<int8_t> cls_of((size_t)n);
  for (int k = 0; k < n; ++k) {
    if (Lt[k] < 1 || (!from_tset && !t_p[k]) || !t_tr_lin[k]) return fail(HHV_E_ARG, "hhv_mac_realign: bad template %d", k);
    cls_of[k] = (int8_t)mac_length_class(Lt[k]);
  }
  for (int k = 0; k < n; ++k) cls.n_long += cls_of[k] == MAC_CLASSES - 1;
  {
    MacClasses cnt = {};
    for (int k = 0; k < n; ++k) cnt.max_Lt[cls_of[k]] = std::max(cnt.max_Lt[cls_of[k]], Lt[k]);
    int max_hits = 0, taken = 0;
    size_t max_lds = 0, lds = 0;
    mac_dataflow_budget(c->num_cus, &max_hits, &max_lds);
    std::vector<int32_t> by_len((size_t)n);
    for (int k = 0; k < n; ++k) by_len[k] = k;
    std::stable_sort(by_len.begin(), by_len.end(), [&](int32_t x, int32_t y) { return Lt[x] > Lt[y]; });
    for (int r = 0; r < n; ++r) {
      const int k = by_len[r], cl = cls_of[k];
      if (cl == MAC_CLASSES - 1) continue;
      const size_t need = mac_rows_lds(cnt.max_Lt[cl], false);
      if (taken < max_hits && lds + need <= max_lds) {
        ++taken;
        lds += need;
      } else {
        cls_of[k] = (int8_t)(MAC_CLASSES - 1);
      }
    }
  }
  for (int k = 0; k < n; ++k) {
    cls.n[cls_of[k]]++;
    cls.max_Lt[cls_of[k]] = std::max(cls.max_Lt[cls_of[k]], Lt[k]);
  }
  std::vector<int32_t> sel((size_t)n);
  {
    int at[MAC_CLASSES] = {};
    for (int cl = 1; cl < MAC_CLASSES; ++cl) at[cl] = at[cl - 1] + cls.n[cl - 1];
    for (int k = 0; k < n; ++k) sel[(size_t)at[cls_of[k]]++] = k;
    // inside a class the longest templates first: workgroups start in this order, and a hit takes a time that grows with its
    // template - the last workgroups to find a CU should be the short ones (longest-processing-time-first)
    int first = 0;
    for (int cl = 0; cl < MAC_CLASSES; ++cl) {
      std::stable_sort(sel.begin() + first, sel.begin() + first + cls.n[cl], [&](int32_t x, int32_t y) { return Lt[x] > Lt[y]; });
      first += cls.n[cl];
    }
  }
  const bool with_ss = c->mac_ss_pending;
  c->mac_ss_pending = false;  // one call only
  if (with_ss) {
    if ((int)c->mac_ss_mode.size() != n || c->mac_ss_Lq != Lq) return fail(HHV_E_ARG, "hhv_mac_realign: hhv_mac_set_ss was called for %zu hits, Lq %d", c->mac_ss_mode.size(), c->mac_ss_Lq);
  }
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
  std::vector<int32_t> vit_i, vit_j, excl_i, excl_j, ends, ranges, res_template;
  bool any_resident = false;
  if (mi) {
    for (int k = 0; k < n; ++k) {
      const hhv_mac_input& h = mi->in[k];
      const bool resident_path = from_tset && !h.i && !h.j;  // Viterbi alignment taken from the set's trace results
      if (resident_path && !mi->ts->hits_valid) {
        delete ms;
        return fail(HHV_E_STATE, "hhv_mac_realign_tset: input %d has no path and the template set has no hhv_hits results", k);
      }
      if (h.nsteps < 0 || h.n_excluded < 0 || (!resident_path && h.nsteps > 0 && (!h.i || !h.j)) ||
          (h.n_excluded > 0 && (!h.excluded_i || !h.excluded_j))) {
        delete ms;
        return fail(HHV_E_ARG, "hhv_mac_realign_hits: bad input %d", k);
      }
      res_template.push_back(resident_path ? mi->template_of[k] : -1);
      any_resident = any_resident || resident_path;
      vit_off[k + 1] = vit_off[k] + (resident_path ? 0 : h.nsteps);
      excl_off[k + 1] = excl_off[k] + h.n_excluded;
    }
    vit_i.resize((size_t)vit_off[n] + 1);
    vit_j.resize((size_t)vit_off[n] + 1);
    excl_i.resize((size_t)excl_off[n] + 1);
    excl_j.resize((size_t)excl_off[n] + 1);
    ends.resize((size_t)n * 4);
    for (int k = 0; k < n; ++k) {
      const hhv_mac_input& h = mi->in[k];
      if (h.nsteps && res_template[k] < 0) {
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
  // One device allocation, carved into 256-byte aligned pieces.  Everything the host hands over comes first and
  // contiguously: it is assembled in a pinned staging buffer of the context with the same layout and goes to the device
  // in ONE copy (two dozen pageable copies of a few bytes to a few megabytes each used to cost as much as the kernels).
  size_t total = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = total;
    total += (bytes + 255) / 256 * 256;
    return at;
  };
  const size_t o_qp = carve((size_t)(Lq + 1) * 20 * 4), o_qtr = carve((size_t)(Lq + 1) * 7 * 4),
               o_tp = carve(from_tset ? (size_t)n * 8 : (size_t)cols * 20 * 4),
               o_ttr = carve((size_t)cols * 7 * 4), o_col = carve((size_t)n * 8), o_Lt = carve((size_t)n * 4),
               o_moff = carve((size_t)n * 8), o_poff = carve((size_t)n * 8);
  const size_t o_ends = carve(ends.size() * 4 + 16), o_voff = carve((size_t)(n + 1) * 8), o_vi = carve(vit_i.size() * 4 + 4),
               o_vj = carve(vit_j.size() * 4 + 4), o_xoff = carve((size_t)(n + 1) * 8), o_xi = carve(excl_i.size() * 4 + 4),
               o_xj = carve(excl_j.size() * 4 + 4), o_rg = carve(ranges.size() * 4), o_rt = carve((size_t)n * 4 + 4);
  const size_t o_sstab = carve(with_ss ? 704 * 4 : 0), o_ssq = carve(with_ss ? c->mac_ss_qidx.size() : 0),
               o_sst = carve(with_ss ? c->mac_ss_tidx.size() : 0), o_ssoff = carve(with_ss ? (size_t)n * 8 : 0),
               o_ssmode = carve(with_ss ? (size_t)n * 4 : 0);
  const size_t o_sel = carve((size_t)n * 4);
  const size_t o_co = carve((size_t)cells);  // explicit masks (hhv_mac_realign) are input too; device-built ones are not
  const size_t input_bytes = mi ? o_co : total;
  const size_t o_mat = carve((size_t)cells * 4), o_bmm = carve((size_t)cells), o_scale = carve((size_t)n * (Lq + 2) * 8),
               o_pf = carve((size_t)n * 8), o_hits = carve((size_t)n * sizeof(DevMacHit)), o_pi = carve((size_t)steps * 4),
               o_pj = carve((size_t)steps * 4), o_ps = carve((size_t)steps), o_pS = carve((size_t)steps * 4),
               o_pP = carve((size_t)steps * 4);
  const size_t path_bytes = total - o_pi;
  const size_t o_rng = carve((size_t)n * (Lq + 2) * 8);  // active column range of every row (hhv_mac_rowrange_kernel)
  const size_t o_rows = carve((size_t)cls.n[MAC_CLASSES - 1] * 10 * (cls.max_Lt[MAC_CLASSES - 1] + 2) * 8);  // row state of the templates beyond LDS
  const bool lists = c->mac_lists;
  const size_t o_fwl = carve(lists ? (size_t)cells * 4 : 0), o_bwl = carve(lists ? (size_t)cells * 4 : 0);
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
  if (c->mac_pinned_bytes < input_bytes) {
    if (c->mac_pinned) (void)hipHostFree(c->mac_pinned);
    c->mac_pinned = nullptr;
    c->mac_pinned_bytes = 0;
    const size_t want = input_bytes + input_bytes / 4;
    if (hipHostMalloc(&c->mac_pinned, want, hipHostMallocDefault) != hipSuccess) {
      hhv_macset_free(ms);
      return fail(HHV_E_MEMORY, "hhv_mac_realign: cannot allocate %zu bytes of pinned host memory", want);
    }
    c->mac_pinned_bytes = want;
  }
  char* base = (char*)ms->d_block;
  char* stage = (char*)c->mac_pinned;
  auto put = [&](size_t at, const void* src, size_t bytes) {
    if (bytes) memcpy(stage + at, src, bytes);
  };
  put(o_qp, q_p, (size_t)(Lq + 1) * 20 * 4);
  put(o_qtr, q_tr_lin, (size_t)(Lq + 1) * 7 * 4);
  for (int k = 0; k < n; ++k) {  // the ragged template operands, packed
    if (from_tset) {
      const int64_t p_off = mi->ts->rec_off[mi->template_of[k]];  // header record of the hit's template in the resident stream
      put(o_tp + (size_t)k * 8, &p_off, 8);
    } else {
      put(o_tp + (size_t)col_off[k] * 20 * 4, t_p[k], (size_t)(Lt[k] + 1) * 20 * 4);
    }
    put(o_ttr + (size_t)col_off[k] * 7 * 4, t_tr_lin[k], (size_t)(Lt[k] + 1) * 7 * 4);
    if (!mi) {
      unsigned char* dst = (unsigned char*)stage + o_co + (size_t)ms->mat_off[k];
      if (celloff && celloff[k]) {
        const uint8_t* src = celloff[k];
        for (size_t e = 0; e < (size_t)(Lq + 1) * (Lt[k] + 1); ++e) dst[e] = src[e] ? 1 : 0;
      } else {
        memset(dst, 0, (size_t)(ms->mat_off[k + 1] - ms->mat_off[k]));
      }
    }
  }
  put(o_col, col_off.data(), (size_t)n * 8);
  put(o_Lt, Lt, (size_t)n * 4);
  put(o_moff, ms->mat_off.data(), (size_t)n * 8);
  put(o_poff, ms->path_off.data(), (size_t)n * 8);
  put(o_sel, sel.data(), (size_t)n * 4);
  if (mi) {
    put(o_ends, ends.data(), ends.size() * 4);
    put(o_voff, vit_off.data(), (size_t)(n + 1) * 8);
    put(o_vi, vit_i.data(), vit_i.size() * 4);
    put(o_vj, vit_j.data(), vit_j.size() * 4);
    put(o_xoff, excl_off.data(), (size_t)(n + 1) * 8);
    put(o_xi, excl_i.data(), excl_i.size() * 4);
    put(o_xj, excl_j.data(), excl_j.size() * 4);
    put(o_rg, ranges.data(), ranges.size() * 4);
    put(o_rt, res_template.data(), (size_t)n * 4);
  }
  if (with_ss) {
    put(o_sstab, c->mac_ss_tab.data(), 704 * 4);
    put(o_ssq, c->mac_ss_qidx.data(), c->mac_ss_qidx.size());
    put(o_sst, c->mac_ss_tidx.data(), c->mac_ss_tidx.size());
    put(o_ssoff, c->mac_ss_toff.data(), (size_t)n * 8);
    put(o_ssmode, c->mac_ss_mode.data(), (size_t)n * 4);
  }
  int rc = HHV_OK;
  hipStream_t st = c->stream;
  if (hipMemcpyAsync(base, stage, input_bytes, hipMemcpyHostToDevice, st) != hipSuccess)
    rc = fail(HHV_E_DEVICE, "hhv_mac_realign: H2D copy failed");
  MacArgs a;
  a.n = n;
  a.Lq = Lq;
  a.q_p = (const float*)(base + o_qp);
  a.q_tr = (const float*)(base + o_qtr);
  a.t_p = from_tset ? mi->ts->d_records : (const float*)(base + o_tp);
  a.t_p_stride = from_tset ? REC_DW : 20;
  a.p_off = from_tset ? (const int64_t*)(base + o_tp) : nullptr;
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
  a.sel = (const int32_t*)(base + o_sel);
  a.row_scratch = (double*)(base + o_rows);
  a.lg2 = c->d_lg2;
  a.diff = c->d_diff;
  a.ss_tab = with_ss ? (const float*)(base + o_sstab) : nullptr;
  a.ss_qidx = with_ss ? (const unsigned char*)(base + o_ssq) : nullptr;
  a.ss_tidx = with_ss ? (const unsigned char*)(base + o_sst) : nullptr;
  a.ss_toff = with_ss ? (const int64_t*)(base + o_ssoff) : nullptr;
  a.ss_mode = with_ss ? (const int32_t*)(base + o_ssmode) : nullptr;
  a.fwd_list = lists ? (float*)(base + o_fwl) : nullptr;
  a.bwd_list = lists ? (float*)(base + o_bwl) : nullptr;
  a.err = c->d_err;
  a.row_rng = (int2*)(base + o_rng);
  {
    // (measurement aid: HHV_MAC_SPARSE_MIN = 0 every hit sparse, a large number none)
    static const int sparse_min = [] { const char* e = getenv("HHV_MAC_SPARSE_MIN"); return e ? atoi(e) : 385; }();
    a.sparse_min_Lt = sparse_min;
    a.ring_min_Lt = 0;
    a.ring_strips = 0;  // (set by launch_mac_class for the single-wave kernels of the longest class)
  }
  ms->d_fwd_list = a.fwd_list;
  ms->d_bwd_list = a.bwd_list;
  // (the kernels write the cells the reference visits: rows 1 .. Lq [- 1], active cells; everything else reads as "no entry")
  if (lists && rc == HHV_OK && hipMemsetAsync(base + o_fwl, 0, (o_bwl - o_fwl) + (size_t)cells * 4, st) != hipSuccess)
    rc = fail(HHV_E_DEVICE, "hhv_mac_realign: memset failed");
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
      m.res_hits = any_resident ? mi->ts->d_hits : nullptr;
      m.res_template = (const int32_t*)(base + o_rt);
      m.res_path_off = any_resident ? mi->ts->d_path_off : nullptr;
      m.res_i = any_resident ? mi->ts->d_i_steps : nullptr;
      m.res_j = any_resident ? mi->ts->d_j_steps : nullptr;
      lr = launch_mac_mask(a, m, st);
    } else {
      lr = launch_mac_rowrange(a, st);
    }
    if (!c->mac_side_ready) {  // the streams the length classes are spread over; without them the classes simply run one after the other
      c->mac_side_ready = true;
      bool ok = hipEventCreateWithFlags((hipEvent_t*)&c->mac_side.fork, hipEventDisableTiming) == hipSuccess;
      for (int k = 0; k < MAC_CLASSES && ok; ++k)  // (created one after the other: the runtime hands out its hardware queues round-robin)
        ok = hipStreamCreateWithFlags((hipStream_t*)&c->mac_side.s[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags((hipEvent_t*)&c->mac_side.join[k], hipEventDisableTiming) == hipSuccess;
      if (!ok)
        for (int k = 0; k < MAC_CLASSES; ++k) c->mac_side.s[k] = nullptr;
    }
    if (lr == 0) lr = launch_mac(a, local != 0, cls, st, c->mac_side.fork ? &c->mac_side : nullptr);
    (void)hipEventRecord(c->ev1, st);
    c->ev_valid = true;
    if (lr != 0) rc = fail(HHV_E_DEVICE, "MAC kernel launch failed: %s", hipGetErrorString((hipError_t)(-lr)));
  }
  static_assert(sizeof(DevMacHit) == sizeof(hhv_mac_hit), "hhv_mac_hit layout");
  ms->hits.resize(n);
  if (c->mac_pinned_out && c->mac_pinned_out_bytes >= path_bytes) {
    ms->h_paths = (char*)c->mac_pinned_out;
    ms->h_paths_bytes = c->mac_pinned_out_bytes;
    c->mac_pinned_out = nullptr;
    c->mac_pinned_out_bytes = 0;
  } else if (rc == HHV_OK) {
    const size_t want = path_bytes + path_bytes / 4;
    if (hipHostMalloc((void**)&ms->h_paths, want, hipHostMallocDefault) != hipSuccess) {
      ms->h_paths = nullptr;
      rc = fail(HHV_E_MEMORY, "hhv_mac_realign: cannot allocate %zu bytes of pinned host memory", want);
    } else {
      ms->h_paths_bytes = want;
    }
  }
  ms->h_pi = 0;
  ms->h_pj = o_pj - o_pi;
  ms->h_ps = o_ps - o_pi;
  ms->h_pS = o_pS - o_pi;
  ms->h_pP = o_pP - o_pi;
  if (rc == HHV_OK && (hipMemcpyAsync(ms->hits.data(), base + o_hits, (size_t)n * sizeof(hhv_mac_hit), hipMemcpyDeviceToHost, st) != hipSuccess ||
                       hipMemcpyAsync(ms->h_paths, base + o_pi, path_bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
                       hipStreamSynchronize(st) != hipSuccess))
    rc = fail(HHV_E_DEVICE, "hhv_mac_realign: kernels failed: %s", hipGetErrorString(hipGetLastError()));
  if (rc == HHV_OK) rc = hhv::api::sync_check(c, "hhv_mac_realign");  // the device error word (a wave of a MAC workgroup that gave up waiting)
  if (rc != HHV_OK) {
    hhv_macset_free(ms);
    return rc;
  }
  memcpy(hits, ms->hits.data(), (size_t)n * sizeof(hhv_mac_hit));
  *out = ms;
  return HHV_OK;
}

int hhv_mac_set_ss(hhv_ctx* c, const float* tables, const uint8_t* q_idx, int32_t Lq, int32_t n, const int32_t* mode,
                   const uint8_t* const* t_idx, const int32_t* Lt) {
  if (!c || !tables || !q_idx || !mode || !t_idx || !Lt || n < 1 || Lq < 1) return fail(HHV_E_ARG, "hhv_mac_set_ss: bad argument");
  c->mac_ss_pending = false;
  c->mac_ss_tab.assign(tables, tables + 2 * 352);
  c->mac_ss_qidx.assign(q_idx, q_idx + (size_t)2 * (Lq + 2));
  c->mac_ss_Lq = Lq;
  c->mac_ss_mode.assign(mode, mode + n);
  c->mac_ss_toff.assign((size_t)n, 0);
  c->mac_ss_tidx.clear();
  for (int k = 0; k < n; ++k) {
    if (mode[k] < 0 || mode[k] > 2 || Lt[k] < 1) return fail(HHV_E_ARG, "hhv_mac_set_ss: hit %d: mode %d, Lt %d", k, mode[k], Lt[k]);
    if (mode[k] && !t_idx[k]) return fail(HHV_E_ARG, "hhv_mac_set_ss: hit %d has a mode but no template indices", k);
    c->mac_ss_toff[k] = (int64_t)c->mac_ss_tidx.size();
    const size_t m = (size_t)Lt[k] + 2;
    if (mode[k]) {
      const int lim = mode[k] == 1 ? 8 : 44;
      for (size_t j = 0; j < m; ++j)
        if (t_idx[k][j] >= lim) return fail(HHV_E_ARG, "hhv_mac_set_ss: hit %d: template index %d at column %zu", k, t_idx[k][j], j);
      c->mac_ss_tidx.insert(c->mac_ss_tidx.end(), t_idx[k], t_idx[k] + m);
    } else {
      c->mac_ss_tidx.insert(c->mac_ss_tidx.end(), m, 0);
    }
  }
  for (int md = 0; md < 2; ++md)
    for (int i = 0; i < Lq + 2; ++i)
      if (q_idx[(size_t)md * (Lq + 2) + i] >= (md == 0 ? 44 : 8)) return fail(HHV_E_ARG, "hhv_mac_set_ss: query index out of range at row %d", i);
  c->mac_ss_pending = true;
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

int hhv_mac_realign_tset(hhv_ctx* c, const float* q_p, const float* q_tr_lin, int32_t Lq, hhv_tset* ts, int32_t n,
                         const int32_t* template_of, const float* const* t_tr_lin, const hhv_mac_input* in, int32_t n_qranges,
                         const int32_t* qranges, int32_t n_tranges, const int32_t* tranges, int32_t local, float shift,
                         float mact, hhv_macset** out, hhv_mac_hit* hits) {
  if (!ts || !template_of || !in || n < 1 || n_qranges < 0 || n_tranges < 0 || (n_qranges && !qranges) || (n_tranges && !tranges))
    return fail(HHV_E_ARG, "hhv_mac_realign_tset: bad argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_mac_realign_tset: template set belongs to another context");
  std::vector<int32_t> Lt(n);
  for (int k = 0; k < n; ++k) {
    if (template_of[k] < 0 || template_of[k] >= ts->n) return fail(HHV_E_ARG, "hhv_mac_realign_tset: template_of[%d] = %d", k, template_of[k]);
    Lt[k] = ts->L[template_of[k]];
  }
  MacMaskInput mi;
  mi.ts = ts;
  mi.template_of = template_of;
  mi.in = in;
  mi.n_qranges = n_qranges;
  mi.qranges = qranges;
  mi.n_tranges = n_tranges;
  mi.tranges = tranges;
  return mac_realign_impl(c, q_p, q_tr_lin, Lq, n, Lt.data(), nullptr, t_tr_lin, nullptr, &mi, local, shift, mact, out, hits);
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
  const char* hp = ms->h_paths;
  if (i_steps) memcpy(i_steps, hp + ms->h_pi + (size_t)o * 4, cnt * 4);
  if (j_steps) memcpy(j_steps, hp + ms->h_pj + (size_t)o * 4, cnt * 4);
  if (states) memcpy(states, hp + ms->h_ps + (size_t)o, cnt);
  if (S) memcpy(S, hp + ms->h_pS + (size_t)o * 4, cnt * 4);
  if (P_posterior) memcpy(P_posterior, hp + ms->h_pP + (size_t)o * 4, cnt * 4);
  return HHV_OK;
}

int hhv_mac_set_lists(hhv_ctx* c, int32_t on) {
  if (!c) return HHV_E_ARG;
  c->mac_lists = on != 0;
  return HHV_OK;
}

// The reference's three sparse lists of a realigned hit (PosteriorDecoder::writeProfilesToHits, src/hhbacktracemac.cpp:14-110),
// in its order (sorted by i, then j - what std::sort with compareIndices leaves; the posterior list is built in that order).
int64_t hhv_mac_list(hhv_macset* ms, int32_t k, int32_t which, int64_t cap, int32_t* li, int32_t* lj, float* lv) {
  if (!ms || k < 0 || k >= ms->n || which < 0 || which > 2 || cap < 0 || (cap > 0 && (!li || !lj || !lv)))
    return fail(HHV_E_ARG, "hhv_mac_list: bad argument");
  if (which < 2 && !ms->d_fwd_list)
    return fail(HHV_E_STATE, "hhv_mac_list: the set was computed without hhv_mac_set_lists(ctx, 1)");
  if (ms->list_k != k || ms->list_which != which) {
    HIP_TRY(hipSetDevice(ms->ctx->par.device));
    const int Lq = ms->Lq, Lt = ms->Lt[k], pitch = Lt + 1;
    const size_t cells = (size_t)(Lq + 1) * pitch;
    std::vector<float> v(cells);
    const float* src = which == 0 ? ms->d_fwd_list : which == 1 ? ms->d_bwd_list : ms->d_mat;
    if (hipMemcpy(v.data(), src + ms->mat_off[k], cells * 4, hipMemcpyDeviceToHost) != hipSuccess)
      return fail(HHV_E_DEVICE, "hhv_mac_list: D2H copy failed");
    std::vector<unsigned char> co;
    if (which == 2) {
      co.resize(cells);
      if (hipMemcpy(co.data(), ms->d_celloff + ms->mat_off[k], cells, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(HHV_E_DEVICE, "hhv_mac_list: D2H copy failed");
      // backtraceMAC has switched off the cells within two rows / columns of every path step by the time the reference builds
      // the list (src/hhbacktracemac.cpp:149-154)
      const int ns = ms->hits[k].nsteps;
      const int32_t* pi = (const int32_t*)(ms->h_paths + ms->h_pi) + ms->path_off[k];
      const int32_t* pj = (const int32_t*)(ms->h_paths + ms->h_pj) + ms->path_off[k];
      for (int s = 1; s <= ns; ++s) {
        const int i = pi[s], j = pj[s];
        for (int ii = std::max(i - 2, 1); ii <= std::min(i + 2, Lq); ++ii) co[(size_t)ii * pitch + j] = 1;
        for (int jj = std::max(j - 2, 1); jj <= std::min(j + 2, Lt); ++jj) co[(size_t)i * pitch + jj] = 1;
      }
    }
    ms->list_k = ms->list_which = -1;
    ms->list_i.clear();
    ms->list_j.clear();
    ms->list_v.clear();
    for (int i = 1; i <= Lq; ++i)
      for (int j = 1; j <= Lt; ++j) {
        const float x = v[(size_t)i * pitch + j];
        bool entry;
        if (which == 2)  // posterior >= POSTERIOR_PROBABILITY_THRESHOLD (src/hhdecl.h:49), cell on, finite (:82-108)
          entry = x >= 0.01f && !co[(size_t)i * pitch + j] && !std::isinf(x) && !std::isnan(x);
        else  // the planes hold the value wherever the reference pushed an entry (a value > 1e-4, possibly inf), 0 elsewhere
          entry = x != 0.0f;
        if (!entry) continue;
        ms->list_i.push_back(i);
        ms->list_j.push_back(j);
        ms->list_v.push_back(x);
      }
    ms->list_k = k;
    ms->list_which = which;
  }
  const int64_t count = (int64_t)ms->list_v.size(), m = std::min(count, cap);
  if (m > 0) {
    memcpy(li, ms->list_i.data(), (size_t)m * sizeof(int32_t));
    memcpy(lj, ms->list_j.data(), (size_t)m * sizeof(int32_t));
    memcpy(lv, ms->list_v.data(), (size_t)m * sizeof(float));
  }
  return count;
}

int hhv_mac_posterior(hhv_macset* ms, int32_t k, float* posterior) {
  if (!ms || k < 0 || k >= ms->n || !posterior) return fail(HHV_E_ARG, "hhv_mac_posterior: bad argument");
  HIP_TRY(hipSetDevice(ms->ctx->par.device));
  HIP_TRY(hipMemcpy(posterior, ms->d_mat + ms->mat_off[k], (size_t)(ms->Lq + 1) * (ms->Lt[k] + 1) * 4, hipMemcpyDeviceToHost));
  return HHV_OK;
}


}  // extern "C"
