// hhv_template_cache.h -- the process-wide resident template cache shared by the drop-in translation units
// (hhviterbirunner_hip.cpp fills it and searches from it; hhposteriordecoderrunner_hip.cpp takes the templates it realigns
// from it instead of parsing them again).  See the header comment of hhviterbirunner_hip.cpp for what is cached and when.
// Header-only: the one instance lives in a function-local static of an inline function.
#ifndef HHV_TEMPLATE_CACHE_H_
#define HHV_TEMPLATE_CACHE_H_

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "hhdatabase.h"
#include "hhdecl.h"
#include "hhhit.h"
#include "util.h"
#include "hhviterbi_hip.h"

namespace hhv_dropin {

// secondary-structure records of a template, [L+1] each, empty when the template has none
struct SsRecords {
  std::vector<int8_t> pred, conf, dssp;
};

// A template of the process-wide cache: raw columns on the device + what a Hit needs to know about the template
// Which database entry a cached template came from: the ffindex data block it lives in and its place there.  Entry name and
// sequence length alone do not identify a template when a process opens several databases (or a rebuilt one).
struct EntryIdentity {
  const void* data;  // FFindexDatabase::db_data of the hhm / a3m / ca3m index that holds the entry (NULL: a file entry)
  uint64_t offset, length;
  bool ambiguous;    // several of the searched databases hold the name: a lookup by name cannot tell which entry this is
  EntryIdentity() : data(NULL), offset(0), length(0), ambiguous(false) {}
  bool operator==(const EntryIdentity& o) const {
    return !ambiguous && !o.ambiguous && data == o.data && offset == o.offset && length == o.length;
  }
};

struct CachedTemplate {
  hhv_rawset* raw;
  int dev;  // slot of TemplateCache::slots whose device holds `raw`
  int32_t index;
  int L;
  EntryIdentity id;
  int ss_pair_mode;
  // The entry is an .hhm TEXT of a database (not an alignment): reading it does not depend on the sequence-weighting
  // argument of getTemplateHMM, so the realign stage (which reads with par.wg, the Viterbi stage with 1) may use it too.
  bool weights_free;
  Hit proto;  // initHitFromHMM(q, t, nseqdis, ssm); its arrays live as long as the cache entry
  SsRecords ss;
  CachedTemplate() : raw(NULL), dev(0), index(0), L(0), ss_pair_mode(0), weights_free(false) {}
};

// One GPU (or one logical shard of a GPU) of the process: its context, whose query it currently holds, its raw sets.
struct DeviceSlot {
  hhv_ctx* ctx;
  int device_id;
  unsigned long owner;  // alignment() call whose query is currently installed in ctx
  std::vector<hhv_rawset*> rawsets;
  size_t columns;
  DeviceSlot() : ctx(NULL), device_id(0), owner(0), columns(0) {}
};

// HHV_DEVICES = "0,1,2,3" / "0-7" / "0,0" (two logical shards on device 0): the devices a search spreads its templates over
// (template-database sharding, SURVEY.md 8e: whole templates, the SIMD batches of the reference are still formed on the whole
// sorted block first, so the results do not depend on the number of devices).  Default: the one device of HHV_DEVICE (0).
inline std::vector<int> configured_devices() {
  std::vector<int> out;
  const char* e = getenv("HHV_DEVICES");
  if (e && *e) {
    const char* p = e;
    while (*p) {
      while (*p && (*p < '0' || *p > '9')) ++p;
      if (!*p) break;
      int a = (int)strtol(p, (char**)&p, 10), b = a;
      if (*p == '-') b = (int)strtol(p + 1, (char**)&p, 10);
      for (int d = a; d <= b && out.size() < 64; ++d) out.push_back(d);
    }
  }
  if (out.empty()) {
    const char* d = getenv("HHV_DEVICE");
    out.push_back(d ? atoi(d) : 0);
  }
  return out;
}

struct TemplateCache {
  std::mutex device;  // one caller at a time on the shared contexts (a search holds it for its device sections)
  std::vector<DeviceSlot> slots;  // slot 0 is the primary device (the realign stage works there)
  unsigned long calls;
  int active;  // searches currently using cache entries
  bool enabled;
  size_t max_columns, columns;
  std::unordered_map<std::string, CachedTemplate> map;
  // what the prototypes depend on besides the template (src/hhhit.cpp:255-256,289-320)
  int nseqdis, ssm, q_has_pred, q_has_dssp;
  // ... and what reading a template from an ALIGNMENT depends on (HHEntry::getTemplateHMM, src/hhdatabase.cpp:299-460): a
  // change empties the cache like a change of nseqdis does
  uint64_t read_param_hash;
  // The background every cached template was read with: HMM::Read overwrites the process-wide `pb` with the NULL line of the file
  // it reads (src/hhhmm.cpp, "NULL" record), and PrepareTemplateHMM runs right behind the read - a template is prepared against ITS
  // OWN file's background, whatever an HMMER-format file read before it (ReadHMMer3 stores the COMPO line in pb, :1399-1404) left
  // there.  Only templates whose NULL line equals this one are cached, and cached templates are always prepared against it - not
  // against whatever the caller's pb holds when the stage starts.
  float null_pb[20];
  bool null_pb_set;
  TemplateCache() : calls(0), active(0), enabled(true), max_columns(0), columns(0), nseqdis(-1), ssm(-1), q_has_pred(-1),
                    q_has_dssp(-1), read_param_hash(0), null_pb_set(false) {
    const char* e = getenv("HHV_TEMPLATE_CACHE");
    enabled = !(e && atoi(e) == 0);
    const char* g = getenv("HHV_TEMPLATE_CACHE_GB");
    const double gb = g ? atof(g) : 64.0;
    max_columns = (size_t)(gb * 1e9 / 128.0);  // 32 dwords per raw column
    const std::vector<int> devs = configured_devices();
    slots.resize(devs.size());
    for (size_t k = 0; k < devs.size(); ++k) slots[k].device_id = devs[k];
  }
  void clear() {  // device lock held
    for (std::unordered_map<std::string, CachedTemplate>::iterator it = map.begin(); it != map.end(); ++it) it->second.proto.Delete();
    map.clear();
    for (size_t d = 0; d < slots.size(); ++d) {
      for (size_t k = 0; k < slots[d].rawsets.size(); ++k) hhv_rawset_free(slots[d].rawsets[k]);
      slots[d].rawsets.clear();
      slots[d].columns = 0;
    }
    columns = 0;
    null_pb_set = false;
  }
};

inline TemplateCache& cache() {
  static TemplateCache c;
  return c;
}

// FNV-1a over the parameters that shape a template read from an a3m / ca3m alignment or with a different size limit
inline uint64_t read_parameters_hash(const Parameters& par, float qsc) {
  uint64_t h = 1469598103934665603ull;
  const int iv[] = {par.M_template, par.Mgaps, par.max_seqid_db, par.coverage_db, par.qid_db, par.Ndiff_db, par.maxres, par.maxcol,
                    par.maxseq, (int)par.mark, (int)par.cons, (int)par.showcons};
  const float fv[] = {qsc, par.qsc_db};
  const unsigned char* b = (const unsigned char*)iv;
  for (size_t k = 0; k < sizeof(iv); ++k) h = (h ^ b[k]) * 1099511628211ull;
  b = (const unsigned char*)fv;
  for (size_t k = 0; k < sizeof(fv); ++k) h = (h ^ b[k]) * 1099511628211ull;
  return h;
}

inline std::string cache_key(HHEntry* e) {
  char len[32];
  snprintf(len, sizeof(len), "\n%d", e->sequence_length);
  return std::string(e->getName()) + len;
}


// The fast_log2 tables THIS PROCESS has (src/util-inl.h:108-130).  They live in function-local statics initialised by the first
// caller, with an initialiser that is compiled per translation unit (double log / logf), so the flavour depends on what ran
// first - an .hhm query or an alignment query, for instance.  lg2[b] is read back through the function itself (x = 1 + b/1024
// has exponent 0 and no low mantissa bits: fast_log2(x) = 0 + lg2[b] + diff[b] * 0); lg2[1024] is 1 in either flavour; diff is
// recomputed from lg2 with the initialiser's own expression.
inline void process_fast_log2_tables(float* lg2, float* diff) {
  for (int b = 0; b < 1024; ++b) {
    const uint32_t bits = 0x3F800000u | ((uint32_t)b << 13);
    float x;
    memcpy(&x, &bits, sizeof(x));
    lg2[b] = fast_log2(x);
  }
  lg2[1024] = 1.0f;
  for (int i = 1; i <= 1024; ++i) diff[i - 1] = (lg2[i] - lg2[i - 1]) * 1.2352E-4;
  diff[1024] = 0.0f;
}

// the shared device contexts (device lock held): created on first use, with the process's own fast_log2 tables
inline int ensure_context(TemplateCache& tc) {
  float lg2[1025], diff[1025];
  bool tables = false;
  for (size_t d = 0; d < tc.slots.size(); ++d) {
    DeviceSlot& sl = tc.slots[d];
    if (sl.ctx) continue;
    hhv_params hp;
    memset(&hp, 0, sizeof(hp));
    hp.device = sl.device_id;
    hp.local = 1;
    int rc = hhv_create(&sl.ctx, &hp);
    if (rc != HHV_OK) return rc;
    if (!tables) process_fast_log2_tables(lg2, diff);
    tables = true;
    rc = hhv_set_fast_log2_tables(sl.ctx, lg2, diff);
    if (rc != HHV_OK) return rc;
  }
  return HHV_OK;
}

// Can PrepareTemplateHMM run on the device for this search?  (hhv_prepare_subset: HHM format, substitution-matrix
// pseudocounts pcm 0..2, null model columnscore 0..3; src/hhfunc.cpp:165-202)
inline bool device_prepare_covers(const Parameters& par) {
  // the library's own definition of what hhv_prepare_subset accepts (pcm 0..3, columnscore 0..3, admixtures that stay in [0, 1]:
  // hhv_prep_params_check, csrc/hhv_api_prep.cpp) - only the fields it looks at are filled
  hhv_prep_params prep;
  memset(&prep, 0, sizeof(prep));
  prep.pcm = par.pc_hhm_nocontext_mode;
  prep.pca = par.pc_hhm_nocontext_a;
  prep.pcb = par.pc_hhm_nocontext_b;
  prep.pcc = par.pc_hhm_nocontext_c;
  prep.columnscore = par.columnscore;
  return hhv_prep_params_check(&prep) == HHV_OK;
}

// the arguments of PrepareTemplateHMM that do not depend on the template
inline hhv_prep_params prepare_params(const Parameters& par, const float* pb, const float R[20][20]) {
  hhv_prep_params prep;
  memset(&prep, 0, sizeof(prep));
  prep.gapd = par.gapd;
  prep.gape = par.gape;
  prep.gapf = par.gapf;
  prep.gapg = par.gapg;
  prep.gaph = par.gaph;
  prep.gapi = par.gapi;
  prep.gapb = par.gapb;
  prep.pcm = par.pc_hhm_nocontext_mode;
  prep.pca = par.pc_hhm_nocontext_a;
  prep.pcb = par.pc_hhm_nocontext_b;
  prep.pcc = par.pc_hhm_nocontext_c;
  prep.columnscore = par.columnscore;
  memcpy(prep.pb, pb, 20 * sizeof(float));
  for (int a = 0; a < 20; ++a)
    for (int b = 0; b < 20; ++b) prep.R[a * 20 + b] = R[a][b];
  return prep;
}

}  // namespace hhv_dropin

#endif  // HHV_TEMPLATE_CACHE_H_
