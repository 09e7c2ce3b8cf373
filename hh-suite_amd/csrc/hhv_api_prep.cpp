// hhv_api_prep.cpp -- C ABI of the on-device PrepareTemplateHMM and the raw template database file.
#include "hhv_api_common.h"

#include <atomic>
#include <chrono>
#include <memory>
#include <thread>

using namespace hhv;
using hhv::api::dfree;
using hhv::api::tfree;
using hhv::api::tmalloc;
using hhv::api::fail;
using hhv::api::tset_init_common;

extern "C" {

// ---- on-device PrepareTemplateHMM (N2) ---------------------------------------------------------------
void hhv_rawset_free(hhv_rawset* rs);
// a float buffer that is not value-initialised (std::vector would zero hundreds of megabytes first)
namespace {
struct RawBlock {
  std::unique_ptr<float[]> p;
  size_t n = 0;
  void resize(size_t m) {
    p.reset(new float[m]);
    n = m;
  }
  float* data() { return p.get(); }
  size_t size() const { return n; }
};
}  // namespace

// one raw HMM -> its L + 1 raw columns of 32 dwords (hhv_internal.h RAW_*) at w
static void fill_raw_template(float* w0, int32_t Lk, const float* f, const float* tr, const float* neff, const int8_t* ss_pred,
                              const int8_t* ss_conf, const int8_t* ss_dssp) {
  for (int i = 0; i <= Lk; ++i) {
    float* w = w0 + (size_t)i * RAW_DW;
    memcpy(w + RAW_F, f + (size_t)i * 20, 20 * sizeof(float));
    memcpy(w + RAW_TR, tr + (size_t)i * 7, 7 * sizeof(float));
    memcpy(w + RAW_NEFF, neff + (size_t)i * 3, 3 * sizeof(float));
    int32_t meta = i;
    if (i >= 1) {
      const int pr = ss_pred ? (unsigned char)ss_pred[i] : 0, cf = ss_conf ? ss_conf[i] : 0;
      const int ds = ss_dssp ? (unsigned char)ss_dssp[i] : 0;
      meta |= (int32_t)((unsigned char)(pr * 11 + cf) & META_PRED_MASK) << META_PRED_SHIFT;
      meta |= (int32_t)(ds & META_DSSP_MASK) << META_DSSP_SHIFT;
    }
    memcpy(w + RAW_J, &meta, 4);
    memcpy(w + RAW_L, &Lk, 4);
  }
}

// raw HMMs -> the 32-dword raw column block the prepare kernels read
static int build_raw_block(int32_t n, const int32_t* L, const float* const* f, const float* const* tr, const float* const* neff,
                           const int8_t* const* ss_pred, const int8_t* const* ss_conf, const int8_t* const* ss_dssp,
                           RawBlock* host) {
  int64_t off = 0;
  for (int k = 0; k < n; ++k) {
    if (L[k] < 1 || L[k] > 0xFFFF || !f[k] || !tr[k] || !neff[k]) return fail(HHV_E_ARG, "raw template %d invalid", k);
    off += (int64_t)L[k] + 1;
  }
  host->resize((size_t)off * RAW_DW);  // every dword of a column block is written below: no zero fill of ~1 GB
  std::vector<int64_t> start((size_t)n + 1, 0);
  for (int k = 0; k < n; ++k) start[k + 1] = start[k] + (int64_t)L[k] + 1;
  auto fill = [&](int k0, int k1) {
    for (int k = k0; k < k1; ++k)
      fill_raw_template(host->data() + (size_t)start[k] * RAW_DW, L[k], f[k], tr[k], neff[k], ss_pred ? ss_pred[k] : nullptr,
                        ss_conf ? ss_conf[k] : nullptr, ss_dssp ? ss_dssp[k] : nullptr);
  };
  // a database upload is hundreds of megabytes of strided copies: spread the templates over a few host threads
  const int nt = (int)std::min<int64_t>(std::min<int64_t>(16, std::max(1u, std::thread::hardware_concurrency())), off / 65536 + 1);
  if (nt <= 1) {
    fill(0, n);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(fill, (int)((int64_t)n * t / nt), (int)((int64_t)n * (t + 1) / nt));
    for (auto& th : pool) th.join();
  }
  return HHV_OK;
}

// Column frequencies are >= 0: what the prepare kernels make of them goes to the DP kernel, whose log2f4 takes the exponent
// of a column product with a plain shift (viterbi_lane.h).  A block from a file of another tool or build is checked like the
// packer checks host profiles: -0 becomes +0, a negative value is refused.  Returns the first bad column of
// [col0, col1) or -1.
static int64_t fix_raw_signs(float* block, size_t col0, size_t col1) {
  int64_t bad = -1;
  for (size_t col = col0; col < col1; ++col) {
    float* fw = block + col * RAW_DW + RAW_F;
    int32_t meta;
    memcpy(&meta, block + col * RAW_DW + RAW_J, 4);
    if ((meta & META_JMASK) == 0) continue;  // row 0 of a template is not a column (nothing reads its frequencies)
    for (int a = 0; a < 20; ++a) {
      uint32_t u;
      memcpy(&u, fw + a, 4);
      if (!(u & 0x80000000u)) continue;
      if (u == 0x80000000u) {
        fw[a] = 0.0f;
      } else if (bad < 0) {
        bad = (int64_t)col;
      }
    }
  }
  return bad;
}

// the resident raw set without its columns: lengths, offsets, class lists, device buffers (d_raw allocated, not filled)
static int rawset_alloc(hhv_ctx* c, int32_t n, const int32_t* L, const float* neff_hmm, size_t block_floats, hhv_rawset** out) {
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
  ok = ok && hipMalloc(&rs->d_raw_off, (size_t)(n + 1) * sizeof(int64_t)) == hipSuccess &&
       hipMemcpy(rs->d_raw_off, rs->rec_off.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && hipMalloc(&rs->d_raw, block_floats * sizeof(float)) == hipSuccess &&
            hipMalloc(&rs->d_neff_hmm, (size_t)n * sizeof(float)) == hipSuccess &&
            hipMalloc(&rs->d_pav, (size_t)n * 20 * sizeof(float)) == hipSuccess &&
            hipMalloc(&rs->d_pb, 20 * sizeof(float)) == hipSuccess && hipMalloc(&rs->d_R, 400 * sizeof(float)) == hipSuccess &&
            hipMalloc(&rs->d_qpav, 20 * sizeof(float)) == hipSuccess &&
            hipMemcpy(rs->d_neff_hmm, neff_hmm, (size_t)n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    hhv_rawset_free(rs);
    return fail(HHV_E_MEMORY, "raw template set: device allocation/copy failed");
  }
  *out = rs;
  return HHV_OK;
}

// raw column block (straight from a raw database file) -> resident raw set
static int rawset_from_block(hhv_ctx* c, int32_t n, const int32_t* L, const float* neff_hmm, float* block,
                             size_t block_floats, hhv_rawset** out) {
  {
    const size_t cols = block_floats / RAW_DW;
    const int nt = (int)std::min<size_t>(std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())), cols / 65536 + 1);
    std::vector<int64_t> bad((size_t)nt, -1);
    auto scan = [&](int w) { bad[(size_t)w] = fix_raw_signs(block, cols * w / nt, cols * (w + 1) / nt); };
    if (nt <= 1) {
      scan(0);
    } else {
      std::vector<std::thread> pool;
      int started = 0;
      try {
        for (; started < nt; ++started) pool.emplace_back(scan, started);
      } catch (...) {
      }
      for (int w = started; w < nt; ++w) scan(w);
      for (auto& th : pool) th.join();
    }
    for (int64_t b : bad)
      if (b >= 0) return fail(HHV_E_ARG, "raw template set: raw column %lld has a negative frequency (f >= 0)", (long long)b);
  }
  hhv_rawset* rs = nullptr;
  const int rc = rawset_alloc(c, n, L, neff_hmm, block_floats, &rs);
  if (rc != HHV_OK) return rc;
  if (hipMemcpy(rs->d_raw, block, block_floats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
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
  const bool timing = getenv("HHV_API_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  int64_t n_cols = 0;
  size_t max_template = 0;
  for (int k = 0; k < n; ++k) {
    if (L[k] < 1 || L[k] > 0xFFFF || !f[k] || !tr[k] || !neff[k]) return fail(HHV_E_ARG, "raw template %d invalid", k);
    n_cols += (int64_t)L[k] + 1;
    max_template = std::max(max_template, (size_t)L[k] + 1);
  }
  hhv_rawset* rs = nullptr;
  int rc = rawset_alloc(c, n, L, neff_hmm, (size_t)n_cols * RAW_DW, &rs);
  if (rc != HHV_OK) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  // Columns packed and uploaded in slabs of <= 64 MiB, as hhv_upload_templates does: two pinned staging buffers used in turn - a
  // slab is packed (and its signs checked) by a few host threads while the copy of the one before it is in flight.  (The whole
  // block packed into pageable memory and one blocking copy: 2 GB/s, 1.9 s for 100 000 templates of 300 columns.)
  const size_t slab_cols = (64u << 20) / (RAW_DW * sizeof(float));
  const size_t stage_floats = std::max(std::min(slab_cols, (size_t)n_cols), max_template) * RAW_DW;
  const int n_stage = (size_t)n_cols > slab_cols ? 2 : 1;
  float* stage[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  auto release = [&]() {
    for (int b = 0; b < 2; ++b) {
      if (done[b]) (void)hipEventDestroy(done[b]);
      if (stage[b]) (void)hipHostFree(stage[b]);
    }
  };
  for (int b = 0; b < n_stage; ++b) {
    if (hipHostMalloc(&stage[b], stage_floats * sizeof(float), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess) {
      release();
      hhv_rawset_free(rs);
      return fail(HHV_E_MEMORY, "hhv_upload_raw_templates: pinned staging of %zu bytes failed", stage_floats * sizeof(float));
    }
  }
  int n_threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("HHV_PACK_THREADS")) n_threads = std::max(1, std::min(64, atoi(e)));
  int k = 0, slab = 0;
  std::atomic<long long> bad_col(-1);
  while (k < n && rc == HHV_OK) {
    const int k0 = k;
    size_t cols = 0;
    while (k < n && (cols == 0 || cols + (size_t)L[k] + 1 <= slab_cols)) {
      cols += (size_t)L[k] + 1;
      ++k;
    }
    const int b = slab & 1;
    if (slab >= 2 && hipEventSynchronize(done[b]) != hipSuccess) {
      rc = fail(HHV_E_DEVICE, "hhv_upload_raw_templates: H2D copy failed");
      break;
    }
    float* const buf = stage[b];
    const int64_t base = rs->rec_off[k0];
    auto pack_range = [&](int t0, int t1) {
      for (int t = t0; t < t1; ++t) {
        float* w = buf + (size_t)(rs->rec_off[t] - base) * RAW_DW;
        fill_raw_template(w, L[t], f[t], tr[t], neff[t], ss_pred ? ss_pred[t] : nullptr, ss_conf ? ss_conf[t] : nullptr,
                          ss_dssp ? ss_dssp[t] : nullptr);
        const int64_t bad = fix_raw_signs(w, 0, (size_t)L[t] + 1);
        if (bad >= 0) {
          long long none = -1;
          bad_col.compare_exchange_strong(none, (long long)(rs->rec_off[t] + bad));
        }
      }
    };
    const int nt = std::max(1, std::min(n_threads, (k - k0) / 64));
    if (nt == 1) {
      pack_range(k0, k);
    } else {
      std::vector<std::thread> pool;
      int started = 0;
      try {
        for (; started < nt; ++started)
          pool.emplace_back(pack_range, k0 + (int)((int64_t)(k - k0) * started / nt), k0 + (int)((int64_t)(k - k0) * (started + 1) / nt));
      } catch (...) {  // no more threads to be had: the calling thread packs the shares that found none
      }
      for (int w = started; w < nt; ++w) pack_range(k0 + (int)((int64_t)(k - k0) * w / nt), k0 + (int)((int64_t)(k - k0) * (w + 1) / nt));
      for (auto& th : pool) th.join();
    }
    if (bad_col.load() >= 0) {
      rc = fail(HHV_E_ARG, "raw template set: raw column %lld has a negative frequency (f >= 0)", bad_col.load());
      break;
    }
    if (hipMemcpyAsync(rs->d_raw + (size_t)base * RAW_DW, buf, cols * RAW_DW * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipEventRecord(done[b], c->stream) != hipSuccess) {
      rc = fail(HHV_E_DEVICE, "hhv_upload_raw_templates: H2D copy failed");
      break;
    }
    ++slab;
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess && rc == HHV_OK) rc = fail(HHV_E_DEVICE, "hhv_upload_raw_templates: H2D copy failed");
  release();
  if (rc != HHV_OK) {
    hhv_rawset_free(rs);
    return rc;
  }
  if (timing)
    fprintf(stderr, "hhv_upload_raw_templates: %d templates, %.1f MB: allocate %.1f ms, pack + copy in %d slabs %.1f ms\n", n,
            (double)n_cols * RAW_DW * 4e-6, std::chrono::duration<double, std::milli>(t1 - t0).count(), slab,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  *out = rs;
  return HHV_OK;
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
  RawBlock host;
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
  dfree(rs->d_raw_off);
  dfree(rs->d_tau);
  delete rs;
}

// pcm 3 recomputes the admixture constant from pcb (src/hhhmm.cpp:1914: a double expression stored in the float pca)
static float prep_pca3(float pcb) { return (float)(0.793 + 0.048 * ((double)pcb - 10.0)); }

static int check_prep_params(const hhv_prep_params* par);
extern "C" int hhv_prep_params_check(const hhv_prep_params* par) {
  if (!par) return fail(HHV_E_ARG, "hhv_prep_params_check: null");
  return check_prep_params(par);
}
static int check_prep_params(const hhv_prep_params* par) {
  if (par->pcm < 0 || par->pcm > 3) return fail(HHV_E_LIMIT, "hhv_prepare_templates: pcm = %d (only 0 .. 3)", par->pcm);
  if (par->columnscore < 0 || par->columnscore > 3)
    return fail(HHV_E_LIMIT, "hhv_prepare_templates: columnscore = %d (only 0..3)", par->columnscore);
  // p = (1 - tau) f + tau g with tau <= pca (src/hhhmm.cpp:1874-1964): an admixture weight above 1 makes profile values
  // negative, which the DP kernel's log2f4 does not take (viterbi_lane.h; the packer refuses them on the host paths too)
  // (pcm 2 clamps tau with fmin(1.0, ..), :1900/:1906: there pca > 1 is legal and only pca >= 0, pcb > 0 are needed - ADVICE r3)
  if (par->pcm == 1 && !(par->pca >= 0.0f && par->pca <= 1.0f))
    return fail(HHV_E_LIMIT, "hhv_prepare_templates: pca = %g; the constant pseudocount admixture (pcm 1) must lie in [0, 1] (profile values stay >= 0)", par->pca);
  if (par->pcm == 3) {
    // :1911-1919: tau = fmax(0, pca3 h(x)), pca3 = 0.793 + 0.048 (pcb - 10), h(x) = 1 - x + pcc x (1 - x), x = Neff_M / pcb >= 0.
    // tau stays in [0, 1] for every column iff pcb > 0, pca3 >= 0 and pca3 * max h <= 1 (max h = 1 for pcc <= 1, else
    // 1 + (pcc - 1)^2 / (4 pcc))
    const float pca3 = prep_pca3(par->pcb);
    const double hmax = par->pcc > 1.0f ? 1.0 + (double)(par->pcc - 1.0f) * (par->pcc - 1.0f) / (4.0 * par->pcc) : 1.0;
    // (<= 0.999, not <= 1: the kernel evaluates pca3 * h(x) in float per column, and a product that is 1 in exact arithmetic may
    // round to 1 + 2^-23 there - a tau above 1 makes (1 - tau) f negative; the reference has no such bound because it keeps
    // negative profile values.  The 0.1 % band is refused here and prepared on the host by the drop-in.  ADVICE r4)
    if (!(par->pcb > 0.0f && pca3 >= 0.0f && par->pcc >= 0.0f && (double)pca3 * hmax <= 0.999))
      return fail(HHV_E_LIMIT, "hhv_prepare_templates: pcm 3 with pcb = %g, pcc = %g can leave [0, 1] with its admixture (profile values stay >= 0)",
                  par->pcb, par->pcc);
  }
  if (par->pcm == 2 && !(par->pca >= 0.0f && par->pcb > 0.0f))
    return fail(HHV_E_LIMIT, "hhv_prepare_templates: pca = %g, pcb = %g; pcm 2 needs pca >= 0 and pcb > 0 (profile values stay >= 0)", par->pca, par->pcb);
  return HHV_OK;
}

// allocates the record stream of a fresh template set and writes its terminal header
static int tset_alloc_stream(hhv_ctx* c, hhv_tset* ts, int32_t n, const int32_t* L) {
  int rc = tset_init_common(c, ts, n, L);
  if (rc == HHV_OK && tmalloc(ts->ctx, &ts->d_records, (size_t)(ts->n_records + STREAM_PAD_RECS) * REC_DW * sizeof(float)) != hipSuccess)
    rc = fail(HHV_E_MEMORY, "hhv_prepare_templates: device allocation failed");
  if (rc != HHV_OK) return rc;
  ts->owns_records = true;
  std::vector<float> tail((size_t)(1 + STREAM_PAD_RECS) * REC_DW, 0.0f);
  write_header(tail.data(), -1, 0);
  HIP_TRY(hipMemcpy(ts->d_records + (size_t)ts->rec_off[n] * REC_DW, tail.data(), tail.size() * sizeof(float), hipMemcpyHostToDevice));
  return HHV_OK;
}

// pcm 2 with pcc != 1: tau[i] = fmin(1, pca / (1 + pow(Neff_M[i] / pcb, pcc))) (src/hhhmm.cpp:1903-1909; float arguments: the
// float overload of pow, i.e. powf).  The device has no bit-exact powf, so the host evaluates it once per raw set and
// (pca, pcb, pcc): Neff_M of every raw column comes back in one strided copy, a few threads call libm, the table goes up.
// This is parameter preparation like the fast_log2 tables, not a fallback of the kernels: the columns are prepared on the device.
static int ensure_tau(hhv_ctx* c, hhv_rawset* rs, const hhv_prep_params* par) {
  if (par->pcm != 2 || par->pcc == 1.0f) return HHV_OK;
  if (rs->d_tau && rs->tau_pca == par->pca && rs->tau_pcb == par->pcb && rs->tau_pcc == par->pcc) return HHV_OK;
  const size_t n = (size_t)rs->n_cols;
  std::vector<float> v(n);
  if (!rs->d_tau) HIP_TRY(hipMalloc(&rs->d_tau, n * sizeof(float)));
  rs->tau_pcc = 1.0f;  // (the table is being rewritten: not valid for any pcc != 1 until it is complete)
  if (const int lr = launch_gather_neff(rs->d_raw, rs->n_cols, rs->d_tau, c->stream); lr != 0)
    return fail(HHV_E_DEVICE, "hhv_prepare_templates: kernel launch failed: %s", hipGetErrorString((hipError_t)(-lr)));
  HIP_TRY(hipMemcpyAsync(v.data(), rs->d_tau, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const float pca = par->pca, pcb = par->pcb, pcc = par->pcc;
  auto eval = [&](size_t a, size_t b) {
    for (size_t i = a; i < b; ++i) v[i] = (float)fmin(1.0, (double)pca / (1. + (double)powf(v[i] / pcb, pcc)));
  };
  const int nt = (int)std::max<size_t>(1, std::min<size_t>(8, n / 65536));
  if (nt == 1) {
    eval(0, n);
  } else {
    std::vector<std::thread> pool;
    int started = 0;
    try {
      for (; started < nt; ++started) pool.emplace_back(eval, n * started / nt, n * (started + 1) / nt);
    } catch (...) {
    }
    for (int w = started; w < nt; ++w) eval(n * w / nt, n * (w + 1) / nt);
    for (auto& th : pool) th.join();
  }
  HIP_TRY(hipMemcpy(rs->d_tau, v.data(), n * sizeof(float), hipMemcpyHostToDevice));
  rs->tau_pca = pca, rs->tau_pcb = pcb, rs->tau_pcc = pcc;
  return HHV_OK;
}

static void fill_prep_args(PrepArgs* a, hhv_ctx* c, hhv_rawset* rs, hhv_tset* ts, const hhv_prep_params* par) {
  a->raw = rs->d_raw;
  a->n_cols = rs->n_cols;
  a->rec_off = ts->d_rec_off;
  a->L = ts->d_L;
  a->neff_hmm = rs->d_neff_hmm;
  a->pb = rs->d_pb;
  a->R = rs->d_R;
  a->q_pav = rs->d_qpav;
  a->lg2 = c->d_lg2;
  a->diff = c->d_diff;
  a->p_tmp = rs->d_p_tmp;
  a->tr_tmp = rs->d_tr_tmp;
  a->records = ts->d_records;
  a->gapd = par->gapd;
  a->gape = par->gape;
  a->gapf = par->gapf;
  a->gapg = par->gapg;
  a->gaph = par->gaph;
  a->gapi = par->gapi;
  a->gapb = par->gapb;
  a->pcm = par->pcm;
  a->pca = par->pcm == 3 ? prep_pca3(par->pcb) : par->pca;
  a->pcb = par->pcb;
  a->pcc = par->pcc;
  a->tau = (par->pcm == 2 && par->pcc != 1.0f) ? rs->d_tau : nullptr;
  a->columnscore = par->columnscore;
  a->ids = nullptr;
  a->lds_cols = 0;
  a->src = nullptr;
  a->raw_off = rs->d_raw_off;
}

int hhv_prepare_templates(hhv_ctx* c, hhv_rawset* rs, const hhv_prep_params* par, const float* q_pav, hhv_tset** out) {
  if (!c || !rs || !par || !q_pav || !out) return fail(HHV_E_ARG, "hhv_prepare_templates: null argument");
  if (rs->ctx != c) return fail(HHV_E_ARG, "hhv_prepare_templates: raw set belongs to another context");
  int rc = check_prep_params(par);
  if (rc != HHV_OK) return rc;
  HIP_TRY(hipSetDevice(c->par.device));
  rc = ensure_tau(c, rs, par);
  if (rc != HHV_OK) return rc;
  hhv_tset* ts = *out;
  if (!ts) {
    ts = new (std::nothrow) hhv_tset();
    if (!ts) return fail(HHV_E_MEMORY, "out of host memory");
    rc = tset_alloc_stream(c, ts, rs->n, rs->L.data());
    if (rc != HHV_OK) {
      hhv_tset_free(ts);
      return rc;
    }
  } else if (ts->n != rs->n || ts->n_records != rs->n_cols + 1) {
    return fail(HHV_E_ARG, "hhv_prepare_templates: *out was not created from this raw set");
  }
  HIP_TRY(hipMemcpyAsync(rs->d_pb, par->pb, 20 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(rs->d_R, par->R, 400 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(rs->d_qpav, q_pav, 20 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  PrepArgs a;
  fill_prep_args(&a, c, rs, ts, par);
  a.pav_out = rs->d_pav;
  (void)hipEventRecord(c->ev0, c->stream);  // hhv_last_kernel_ms: the prepare kernels of this call
  rc = launch_prepare(a, rs->d_ids, rs->n_ids, rs->max_L, c->stream);
  (void)hipEventRecord(c->ev1, c->stream);
  c->ev_valid = true;
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

int hhv_prepare_subset(hhv_ctx* c, hhv_rawset* rs, const hhv_prep_params* par, const float* q_pav, const int32_t* ids,
                       int32_t n_ids, hhv_tset** out) {
  if (!c || !rs || !par || !q_pav || !ids || !out) return fail(HHV_E_ARG, "hhv_prepare_subset: null argument");
  if (rs->ctx != c) return fail(HHV_E_ARG, "hhv_prepare_subset: raw set belongs to another context");
  if (n_ids < 1) return fail(HHV_E_ARG, "hhv_prepare_subset: n_ids = %d", n_ids);
  *out = nullptr;
  int rc = check_prep_params(par);
  if (rc != HHV_OK) return rc;
  std::vector<int32_t> L(n_ids), cls[3];
  int32_t max_L[3] = {0, 0, 0}, n_cls[3];
  for (int k = 0; k < n_ids; ++k) {
    if (ids[k] < 0 || ids[k] >= rs->n) return fail(HHV_E_ARG, "hhv_prepare_subset: ids[%d] = %d of %d", k, ids[k], rs->n);
    L[k] = rs->L[ids[k]];
    const int cl = L[k] <= 447 ? 0 : (L[k] <= 1300 ? 1 : 2);
    cls[cl].push_back(k);
    max_L[cl] = std::max(max_L[cl], L[k]);
  }
  HIP_TRY(hipSetDevice(c->par.device));
  rc = ensure_tau(c, rs, par);
  if (rc != HHV_OK) return rc;
  if (!cls[2].empty() && !rs->d_p_tmp) {  // the raw set had no template this long when it was uploaded?  cannot happen:
    return fail(HHV_E_STATE, "hhv_prepare_subset: intermediate buffers missing");  // the classes depend on L only
  }
  hhv_tset* ts = new (std::nothrow) hhv_tset();
  if (!ts) return fail(HHV_E_MEMORY, "out of host memory");
  rc = tset_alloc_stream(c, ts, n_ids, L.data());
  int32_t* d_scratch = nullptr;  // src[n_ids] followed by the three slot lists
  if (rc == HHV_OK && tmalloc(c, &d_scratch, (size_t)2 * n_ids * sizeof(int32_t)) != hipSuccess)
    rc = fail(HHV_E_MEMORY, "hhv_prepare_subset: device allocation failed");
  const int32_t* d_cls[3] = {nullptr, nullptr, nullptr};
  if (rc == HHV_OK) {
    std::vector<int32_t> lists;
    lists.reserve(n_ids);
    size_t at = n_ids;
    for (int cl = 0; cl < 3; ++cl) {
      n_cls[cl] = (int32_t)cls[cl].size();
      d_cls[cl] = d_scratch + at;
      at += cls[cl].size();
      lists.insert(lists.end(), cls[cl].begin(), cls[cl].end());
    }
    if (hipMemcpyAsync(d_scratch, ids, (size_t)n_ids * sizeof(int32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_scratch + n_ids, lists.data(), (size_t)n_ids * sizeof(int32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(rs->d_pb, par->pb, 20 * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(rs->d_R, par->R, 400 * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(rs->d_qpav, q_pav, 20 * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess)
      rc = fail(HHV_E_DEVICE, "hhv_prepare_subset: H2D copy failed");
  }
  if (rc == HHV_OK) {
    PrepArgs a;
    fill_prep_args(&a, c, rs, ts, par);
    a.src = d_scratch;
    a.pav_out = nullptr;
    (void)hipEventRecord(c->ev0, c->stream);
    const int lr = launch_prepare(a, d_cls, n_cls, max_L, c->stream);
    (void)hipEventRecord(c->ev1, c->stream);
    c->ev_valid = true;
    if (lr != 0) rc = fail(HHV_E_DEVICE, "prepare kernel launch failed: %s", hipGetErrorString((hipError_t)(-lr)));
  }
  if (rc == HHV_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(HHV_E_DEVICE, "hhv_prepare_subset: kernels failed");
  tfree(c, d_scratch);
  if (rc != HHV_OK) {
    hhv_tset_free(ts);
    return rc;
  }
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

int hhv_tset_download(hhv_ctx* c, hhv_tset* ts, float* out) {
  if (!c || !ts || !out) return fail(HHV_E_ARG, "hhv_tset_download: null argument");
  HIP_TRY(hipSetDevice(c->par.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(out, ts->d_records, (size_t)ts->n_records * REC_DW * sizeof(float), hipMemcpyDeviceToHost));
  return HHV_OK;
}

}  // extern "C"
