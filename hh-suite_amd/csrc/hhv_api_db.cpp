// hhv_api_db.cpp -- C ABI of the binary packed template database.
#include "hhv_api_common.h"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>

using namespace hhv;
using hhv::api::dfree;
using hhv::api::tfree;
using hhv::api::tmalloc;
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
  if (rc == HHV_OK && tmalloc(ts->ctx, &ts->d_records, (size_t)(ts->n_records + STREAM_PAD_RECS) * REC_DW * sizeof(float)) != hipSuccess)
    rc = fail(HHV_E_MEMORY, "hhv_db_open: device allocation failed");
  if (rc == HHV_OK) {
    ts->owns_records = true;
    (void)hipMemset(ts->d_records + (size_t)ts->n_records * REC_DW, 0, (size_t)STREAM_PAD_RECS * REC_DW * sizeof(float));
    // The kernel takes the template index and the template boundaries from the records themselves (it writes
    // results[index] and resets at meta < 0), so a stale or damaged file must not reach the device: every record's meta
    // word is checked against the length table while the slabs go by.  Slabs of 64 MiB, two pinned buffers used in turn: a
    // few host threads read (pread, page cache -> pinned memory) and check their share of a slab while the copy of the slab
    // before it is in flight.
    const int64_t slab_recs = (int64_t)((64u << 20) / (REC_DW * sizeof(float)));
    const int64_t data0 = (int64_t)sizeof(DbHeader) + (int64_t)h.n * 4;
    const int fd = fileno(f);
    float* stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int b = 0; b < 2 && rc == HHV_OK; ++b) {
      if (hipHostMalloc(&stage[b], (size_t)slab_recs * REC_DW * sizeof(float), hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess)
        rc = fail(HHV_E_MEMORY, "hhv_db_open: pinned staging failed");
    }
    int n_threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("HHV_PACK_THREADS")) n_threads = std::max(1, std::min(64, atoi(e)));
    struct Bad {
      int64_t rec = -1;  // first bad record of a thread's share (-1: none); kind 0 = short read, 1 = header, 2 = column
      int kind = 0, tmpl = 0, j = 0;
    };
    const std::vector<int64_t>& ro = ts->rec_off;  // [n + 1]: header record of template k; ro[n] = the terminal header
    int slab = 0;
    for (int64_t r0 = 0; r0 < ts->n_records && rc == HHV_OK; r0 += slab_recs, ++slab) {
      const int64_t r1 = std::min<int64_t>(ts->n_records, r0 + slab_recs);
      const int b = slab & 1;
      if (slab >= 2 && hipEventSynchronize(done[b]) != hipSuccess) {
        rc = fail(HHV_E_DEVICE, "hhv_db_open: H2D copy failed");
        break;
      }
      float* const buf = stage[b];
      const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, (r1 - r0) / 4096));
      std::vector<Bad> bad((size_t)nt);
      auto share = [&](int w) {
        const int64_t a0 = r0 + (r1 - r0) * w / nt, a1 = r0 + (r1 - r0) * (w + 1) / nt;
        Bad& bd = bad[(size_t)w];
        size_t want = (size_t)(a1 - a0) * REC_DW * sizeof(float), got = 0;
        char* dst = reinterpret_cast<char*>(buf + (size_t)(a0 - r0) * REC_DW);
        while (got < want) {
          const ssize_t m = pread(fd, dst + got, want - got, (off_t)(data0 + a0 * REC_DW * (int64_t)sizeof(float) + (int64_t)got));
          if (m <= 0) {
            bd.rec = a0;
            bd.kind = 0;
            return;
          }
          got += (size_t)m;
        }
        // template of record a0: the last header at or before it (the terminal header counts as template n)
        int32_t tmpl = (int32_t)(std::upper_bound(ro.begin(), ro.end(), a0) - ro.begin()) - 1;
        for (int64_t rec = a0; rec < a1; ++rec) {
          float* r = buf + (size_t)(rec - r0) * REC_DW;
          int32_t meta, w0, w1;
          memcpy(&meta, r + REC_META, 4);
          memcpy(&w0, r, 4);
          memcpy(&w1, r + 1, 4);
          if (tmpl < h.n && rec == ro[(size_t)tmpl + 1]) ++tmpl;
          if (rec == ro[(size_t)tmpl]) {  // header of template `tmpl` (or the terminal header)
            const bool last = tmpl == h.n;
            if (meta >= 0 || (last ? w0 != -1 : (w0 != tmpl || w1 != L[(size_t)tmpl]))) {
              bd.rec = rec, bd.kind = 1, bd.tmpl = tmpl;
              return;
            }
          } else {
            const int32_t j = (int32_t)(rec - ro[(size_t)tmpl]);
            if (meta < 0 || (meta & META_JMASK) != j || (((meta & META_LAST) != 0) != (j == L[(size_t)tmpl]))) {
              bd.rec = rec, bd.kind = 2, bd.tmpl = tmpl, bd.j = j;
              return;
            }
            // the kernel takes the exponent of a column product with a plain shift (viterbi_lane.h log2f4): a profile word
            // with the sign bit set - a file written by another tool or an older build - would give a wrong score silently.
            // -0 becomes +0 (same value in every product), anything negative is refused like the packer refuses it
            for (int a = 0; a < 20; ++a) {
              uint32_t u;
              memcpy(&u, r + a, 4);
              if (u & 0x80000000u) {
                if (u == 0x80000000u) {
                  r[a] = 0.0f;
                } else {
                  bd.rec = rec, bd.kind = 3, bd.tmpl = tmpl, bd.j = j;
                  return;
                }
              }
            }
          }
        }
      };
      if (nt == 1) {
        share(0);
      } else {
        std::vector<std::thread> pool;
        int started = 0;
        try {
          for (; started < nt; ++started) pool.emplace_back(share, started);
        } catch (...) {  // no more threads to be had: the calling thread does the shares that found none
        }
        for (int w = started; w < nt; ++w) share(w);
        for (auto& th : pool) th.join();
      }
      for (const Bad& bd : bad) {  // shares are in record order: the first one that failed holds the first bad record
        if (bd.rec < 0) continue;
        if (bd.kind == 0) rc = fail(HHV_E_ARG, "hhv_db_open: truncated record stream");
        else if (bd.kind == 1)
          rc = fail(HHV_E_ARG, "hhv_db_open: %s: record %lld is not the header of template %d", path, (long long)bd.rec, bd.tmpl);
        else if (bd.kind == 3)
          rc = fail(HHV_E_ARG, "hhv_db_open: %s: column %d of template %d has a negative profile value (p = f / null model >= 0)", path, bd.j, bd.tmpl);
        else
          rc = fail(HHV_E_ARG, "hhv_db_open: %s: record %lld is not column %d of template %d", path, (long long)bd.rec, bd.j, bd.tmpl);
        break;
      }
      if (rc != HHV_OK) break;
      if (hipMemcpyAsync(ts->d_records + (size_t)r0 * REC_DW, buf, (size_t)(r1 - r0) * REC_DW * sizeof(float), hipMemcpyHostToDevice,
                         c->stream) != hipSuccess ||
          hipEventRecord(done[b], c->stream) != hipSuccess)
        rc = fail(HHV_E_DEVICE, "hhv_db_open: H2D copy failed");
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == HHV_OK) rc = fail(HHV_E_DEVICE, "hhv_db_open: H2D copy failed");
    for (int b = 0; b < 2; ++b) {
      if (done[b]) (void)hipEventDestroy(done[b]);
      if (stage[b]) (void)hipHostFree(stage[b]);
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
  if (rc == HHV_OK && (tmalloc(c, &sub->d_records, (size_t)(sub->n_records + STREAM_PAD_RECS) * REC_DW * sizeof(float)) != hipSuccess ||
                       tmalloc(c, &d_ids, (size_t)n * sizeof(int32_t)) != hipSuccess))
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
  tfree(c, d_ids);
  if (rc != HHV_OK) {
    hhv_tset_free(sub);
    return rc;
  }
  *out = sub;
  return HHV_OK;
}

}  // extern "C"
