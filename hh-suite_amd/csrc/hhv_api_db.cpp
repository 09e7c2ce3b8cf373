// hhv_api_db.cpp -- C ABI of the binary packed template database.
#include "hhv_api_common.h"

#include <sys/stat.h>

using namespace hhv;
using hhv::api::dfree;
using hhv::api::fail;
using hhv::api::tset_init_common;

extern "C" {

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
  int negative = -1;
  for (int k = 0; k < n && ok; ++k) {
    buf.resize(((size_t)L[k] + 1) * REC_DW);
    if (!pack_template(p[k], tr[k], L[k], k, buf.data(), ss_pred ? ss_pred[k] : nullptr, ss_conf ? ss_conf[k] : nullptr,
                       ss_dssp ? ss_dssp[k] : nullptr)) {
      negative = k;
      break;
    }
    ok = fwrite(buf.data(), sizeof(float), buf.size(), f) == buf.size();
  }
  if (negative >= 0) {
    fclose(f);
    remove(path);
    return fail(HHV_E_ARG, "hhv_db_write: template %d has a negative profile value", negative);
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
  // the header is checked against the file size before anything is allocated from it
  struct stat st;
  if (fstat(fileno(f), &st) != 0 || h.n_records < (int64_t)h.n * 2 + 1 ||
      (int64_t)st.st_size != (int64_t)sizeof(DbHeader) + (int64_t)h.n * 4 + h.n_records * REC_DW * 4) {
    fclose(f);
    return fail(HHV_E_ARG, "hhv_db_open: %s: header (%d templates, %lld records) does not match the file size", path, h.n,
                (long long)h.n_records);
  }
  std::vector<int32_t> L((size_t)h.n);
  if (fread(L.data(), sizeof(int32_t), L.size(), f) != L.size()) {
    fclose(f);
    return fail(HHV_E_ARG, "hhv_db_open: truncated length table");
  }
  if (const hipError_t e = hipSetDevice(c->par.device); e != hipSuccess) {
    fclose(f);
    return fail(HHV_E_DEVICE, "hhv_db_open: hipSetDevice(%d): %s", c->par.device, hipGetErrorString(e));
  }
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
    const size_t slab = (64u << 20) / sizeof(float) / REC_DW * REC_DW;  // whole records per slab
    std::vector<float> buf(slab);
    size_t left = (size_t)ts->n_records * REC_DW, off = 0;
    // The kernel takes the template index and the template boundaries from the records themselves (it writes
    // results[index] and resets at meta < 0), so a stale or damaged file must not reach the device: every record's meta
    // word is checked against the length table while the slabs go by.
    int64_t rec = 0;   // stream record index
    int32_t tmpl = 0;  // template the record belongs to
    int64_t next_hdr = 0;
    while (left && rc == HHV_OK) {
      const size_t m = std::min(left, slab);
      if (fread(buf.data(), sizeof(float), m, f) != m) rc = fail(HHV_E_ARG, "hhv_db_open: truncated record stream");
      for (size_t r0 = 0; r0 < m && rc == HHV_OK; r0 += REC_DW, ++rec) {
        int32_t meta, w0, w1;
        memcpy(&meta, &buf[r0 + REC_META], 4);
        memcpy(&w0, &buf[r0], 4);
        memcpy(&w1, &buf[r0 + 1], 4);
        if (rec == next_hdr) {  // header of template `tmpl` (or the terminal header)
          const bool last = tmpl == h.n;
          if (meta >= 0 || (last ? w0 != -1 : (w0 != tmpl || w1 != L[tmpl])))
            rc = fail(HHV_E_ARG, "hhv_db_open: %s: record %lld is not the header of template %d", path, (long long)rec, tmpl);
          if (!last) next_hdr = rec + L[tmpl] + 1;
          ++tmpl;
        } else {
          const int32_t j = (int32_t)(rec - (next_hdr - L[tmpl - 1] - 1));
          if (meta < 0 || (meta & META_JMASK) != j || (((meta & META_LAST) != 0) != (j == L[tmpl - 1])))
            rc = fail(HHV_E_ARG, "hhv_db_open: %s: record %lld is not column %d of template %d", path, (long long)rec, j, tmpl - 1);
        }
      }
      if (rc != HHV_OK) break;
      if (hipMemcpy(ts->d_records + off, buf.data(), m * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
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

// A new template set holding templates ids[0..n) of a resident one (any order, repeats allowed), copied on the device.
int hhv_tset_gather(hhv_ctx* c, hhv_tset* ts, const int32_t* ids, int32_t n, hhv_tset** out) {
  if (!c || !ts || !ids || !out || n < 1) return fail(HHV_E_ARG, "hhv_tset_gather: bad argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_tset_gather: template set belongs to another context");
  *out = nullptr;
  std::vector<int32_t> L(n);
  for (int k = 0; k < n; ++k) {
    if (ids[k] < 0 || ids[k] >= ts->n) return fail(HHV_E_ARG, "hhv_tset_gather: ids[%d] = %d of %d", k, ids[k], ts->n);
    L[k] = ts->L[ids[k]];
  }
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_tset* sub = new (std::nothrow) hhv_tset();
  if (!sub) return fail(HHV_E_MEMORY, "out of host memory");
  int rc = tset_init_common(c, sub, n, L.data());
  int32_t* d_ids = nullptr;
  if (rc == HHV_OK && (hipMalloc(&sub->d_records, (size_t)(sub->n_records + STREAM_PAD_RECS) * REC_DW * sizeof(float)) != hipSuccess ||
                       hipMalloc(&d_ids, (size_t)n * sizeof(int32_t)) != hipSuccess))
    rc = fail(HHV_E_MEMORY, "hhv_tset_gather: device allocation failed");
  if (rc == HHV_OK) {
    sub->owns_records = true;
    std::vector<float> tail((size_t)(1 + STREAM_PAD_RECS) * REC_DW, 0.0f);
    write_header(tail.data(), -1, 0);
    if (hipMemcpyAsync(sub->d_records + (size_t)sub->rec_off[n] * REC_DW, tail.data(), tail.size() * sizeof(float),
                       hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_ids, ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess)
      rc = fail(HHV_E_DEVICE, "hhv_tset_gather: H2D copy failed");
  }
  if (rc == HHV_OK && tset_gather(ts->d_records, ts->d_rec_off, d_ids, sub->d_rec_off, sub->d_L, n, sub->d_records, c->stream) != 0)
    rc = fail(HHV_E_DEVICE, "hhv_tset_gather: kernel launch failed");
  if (rc == HHV_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(HHV_E_DEVICE, "hhv_tset_gather: kernel failed");
  dfree(d_ids);
  if (rc != HHV_OK) {
    hhv_tset_free(sub);
    return rc;
  }
  *out = sub;
  return HHV_OK;
}

}  // extern "C"
