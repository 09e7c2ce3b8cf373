// hhprefilter_hip.cpp -- DROP-IN replacement for src/hhprefilter.cpp of hh-suite v3.3.0.
//
// Same class Prefilter as declared by the reference's own src/hhprefilter.h (untouched): constructor, destructor,
// init_no_prefiltering, init_selected and prefilter_db keep their signatures, HHblitsDatabase keeps calling them
// (src/hhdatabase.cpp:128-190).  A maintainer compiles this file instead of src/hhprefilter.cpp and links
// libhhv_runner.so + libhhviterbi_hip.so.
//
// What moves to the GPU: the two scans of prefilter_db (src/hhprefilter.cpp:461-530) - the gapless profile/sequence
// score of EVERY column-state sequence of the database with the first selection (length correction, sort, cut) and the
// striped Smith-Waterman of the survivors - on a copy of the cs219 database that is uploaded once per Prefilter object
// and stays in HBM (hhv::Prefilter above hhv_prefilter_upload_db / hhv_prefilter_first / hhv_prefilter_scores).  What
// stays here: reading the context library and the database index (the reference's own cs / ffindex code), the e-value
// selection of the second stage on the <= min_prefilter_hits..n survivors (hhv::Prefilter::SelectSecond reproduces the
// order of the reference's sort, including its int-truncated keys), and the name handling at the end (:555-590).
// The scores are those of the reference's AVX2 build (32-byte stripes), tests/test_prefilter.py.
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "hhprefilter.h"
#ifdef HHV_DROPIN_SHARED_RESOURCE  // test build: the reference's own hhprefilter.o is linked as well and defines the resource
extern unsigned char cs219_lib[];
extern unsigned int cs219_lib_len;
#else
#include "cs219.lib.h"
#endif
#include "hhviterbi_hip.h"
// hh-suite_amd/host/prefilter.h: hhv::Prefilter, the C++ class above the C ABI.  (The test build renames the reference's
// class Prefilter with a -D macro so that both implementations fit into one process; the macro must not touch hhv::Prefilter.)
#pragma push_macro("Prefilter")
#undef Prefilter
#include "prefilter.h"
typedef hhv::Prefilter DevicePrefilter;
#pragma pop_macro("Prefilter")

namespace {

// the class layout belongs to the reference's header, so the device side of an object lives beside it
struct DeviceSide {
  hhv_ctx* ctx;
  DevicePrefilter* pf;
  std::mutex* busy;  // hhblits_omp: several threads share one Prefilter object; its device context serves one at a time
  DeviceSide() : ctx(NULL), pf(NULL), busy(NULL) {}
};
std::mutex g_side_mutex;
std::map<const Prefilter*, DeviceSide> g_side;

void pf_check(int rc, const char* what) {
  if (rc == HHV_OK) return;
  HH_LOG(ERROR) << "hhviterbi_hip: " << what << " failed: " << hhv_last_error() << std::endl;
  exit(rc == HHV_E_MEMORY ? 3 : 4);
}

}  // namespace

Prefilter::Prefilter(const std::string& cs_library, FFindexDatabase* cs219_database) {
  num_dbs = 0;
  // the column-state library, read by the reference's cs code (src/hhprefilter.cpp:31-44)
  FILE* fin;
  if (cs_library.empty()) fin = fmemopen((void*)cs219_lib, cs219_lib_len, "r");
  else fin = fopen(cs_library.c_str(), "r");
  if (!fin) OpenFileError(cs_library.c_str(), __FILE__, __LINE__, __func__);
  cs_lib = new cs::ContextLibrary<cs::AA>(fin);
  fclose(fin);
  cs::TransformToLin(*cs_lib);
  init_prefilter(cs219_database);
}

Prefilter::~Prefilter() {
  {
    std::lock_guard<std::mutex> lock(g_side_mutex);
    std::map<const Prefilter*, DeviceSide>::iterator it = g_side.find(this);
    if (it != g_side.end()) {
      delete it->second.pf;
      hhv_destroy(it->second.ctx);
      delete it->second.busy;
      g_side.erase(it);
    }
  }
  free(length);
  free(first);
  for (size_t n = 0; n < num_dbs; n++) delete[] dbnames[n];
  free(dbnames);
  delete cs_lib;
}

namespace {
// (length, name) of one index entry, the pair HHblitsDatabase::getEntriesFromNames expects
std::pair<int, std::string> length_and_name(const ffindex_entry_t* e) { return std::make_pair((int)e->length, std::string(e->name)); }
}  // namespace

// no prefiltering (hhsearch): every entry of the database, in index order (src/hhprefilter.cpp:280-294)
void Prefilter::init_no_prefiltering(FFindexDatabase* query_database, std::vector<std::pair<int, std::string> >& prefiltered_entries) {
  ffindex_index_t* index = query_database->db_index;
  prefiltered_entries.reserve(prefiltered_entries.size() + index->n_entries);
  for (size_t k = 0; k < index->n_entries; ++k) prefiltered_entries.push_back(length_and_name(ffindex_get_entry_by_index(index, k)));
  HH_LOG(INFO) << "Searching " << prefiltered_entries.size() << " database HHMs without prefiltering" << std::endl;
}

// the entries named by the caller, in the caller's order (src/hhprefilter.cpp:296-310)
void Prefilter::init_selected(FFindexDatabase* cs219_database, std::vector<std::string> templates,
                              std::vector<std::pair<int, std::string> >& prefiltered_entries) {
  for (std::vector<std::string>::iterator name = templates.begin(); name != templates.end(); ++name)
    prefiltered_entries.push_back(length_and_name(ffindex_get_entry_by_name(cs219_database->db_index, const_cast<char*>(name->c_str()))));
}

// src/hhprefilter.cpp:315-338 + the upload: the column-state sequences go to the device once and stay there
void Prefilter::init_prefilter(FFindexDatabase* cs219_database) {
  num_dbs = cs219_database->db_index->n_entries;
  first = (unsigned char**)mem_align(ALIGN_FLOAT, num_dbs * sizeof(unsigned char*));
  length = (int*)mem_align(ALIGN_FLOAT, num_dbs * sizeof(int));
  dbnames = (char**)mem_align(ALIGN_FLOAT, num_dbs * sizeof(char*));
  std::vector<int64_t> offsets(num_dbs + 1, 0);
  for (size_t n = 0; n < num_dbs; n++) {
    ffindex_entry_t* entry = ffindex_get_entry_by_index(cs219_database->db_index, n);
    first[n] = (unsigned char*)ffindex_get_data_by_entry(cs219_database->db_data, entry);
    length[n] = entry->length - 1;
    dbnames[n] = new char[strlen(entry->name) + 1];
    strcpy(dbnames[n], entry->name);
    offsets[n + 1] = offsets[n] + length[n];
  }
  checkCSFormat(5);
  std::vector<uint8_t> seqs((size_t)offsets[num_dbs]);
  for (size_t n = 0; n < num_dbs; n++) memcpy(seqs.data() + offsets[n], first[n], (size_t)length[n]);
  std::vector<double> lib((size_t)cs::AS219::kSize * 20);
  for (size_t k = 0; k < cs::AS219::kSize; ++k)
    for (int a = 0; a < 20; ++a) lib[k * 20 + a] = (*cs_lib)[k].probs[0][a];

  DeviceSide side;
  hhv_params hp;
  memset(&hp, 0, sizeof(hp));
  const char* dev = getenv("HHV_DEVICE");
  hp.device = dev ? atoi(dev) : 0;
  hp.local = 1;
  pf_check(hhv_create(&side.ctx, &hp), "hhv_create");
  side.busy = new std::mutex();
  side.pf = new DevicePrefilter(side.ctx, (int32_t)num_dbs, seqs.data(), offsets.data(), lib.data());
  if (!side.pf->ok()) pf_check(HHV_E_DEVICE, "hhv_prefilter_upload_db");
  {
    std::lock_guard<std::mutex> lock(g_side_mutex);
    g_side[this] = side;
  }
  HH_LOG(INFO) << "Searching " << num_dbs << " column state sequences." << std::endl;
}

// The old text format of cs219 databases starts every entry with '>': refused like the reference does
// (src/hhprefilter.cpp:340-353, including its way of counting: the limit of the scan shrinks with every hit)
void Prefilter::checkCSFormat(size_t nr_checks) {
  size_t remaining = nr_checks;
  for (size_t n = 0; n < std::min(remaining, num_dbs); n++)
    if (first[n][0] == '>') remaining--;
  if (remaining == 0) {
    HH_LOG(ERROR) << "hhprefilter: the column-state database is in the old text format, which is no longer supported "
                     "(see the user manual)" << std::endl;
    exit(1);
  }
}

void Prefilter::prefilter_db(HMM* q_tmp, Hash<Hit>* previous_hits, const int threads, const int prefilter_gap_open,
                             const int prefilter_gap_extend, const int prefilter_score_offset, const int prefilter_bit_factor,
                             const double prefilter_evalue_thresh, const double prefilter_evalue_coarse_thresh,
                             const int preprefilter_smax_thresh, const int min_prefilter_hits, const int maxnumdb,
                             const float R[20][20], std::vector<std::pair<int, std::string> >& new_prefilter_hits,
                             std::vector<std::pair<int, std::string> >& old_prefilter_hits) {
  DeviceSide side;
  {
    std::lock_guard<std::mutex> lock(g_side_mutex);
    side = g_side[this];
  }
  hhv::PrefilterParams pp;
  pp.gap_open = prefilter_gap_open;
  pp.gap_extend = prefilter_gap_extend;
  pp.score_offset = prefilter_score_offset;
  pp.bit_factor = prefilter_bit_factor;
  pp.evalue_thresh = prefilter_evalue_thresh;
  pp.evalue_coarse_thresh = prefilter_evalue_coarse_thresh;
  pp.smax_thresh = preprefilter_smax_thresh;
  pp.min_hits = min_prefilter_hits;
  pp.maxnumdb = maxnumdb;
  const int LQ = q_tmp->L;
  std::vector<float> qp((size_t)LQ * 20);  // the rows stripe_query_profile reads: p[0..LQ-1] (:364-370)
  for (int i = 0; i < LQ; ++i) memcpy(&qp[(size_t)i * 20], q_tmp->p[i], 20 * sizeof(float));
  std::vector<int32_t> ids;
  int passed_first = 0;
  {
    std::lock_guard<std::mutex> one_at_a_time(*side.busy);
    pf_check(side.pf->prefilter_db(qp.data(), q_tmp->pav, LQ, pp, &ids, NULL, &passed_first), "prefilter_db");
  }
  HH_LOG(INFO) << "HMMs passed 1st prefilter (gapless profile-profile alignment)  : " << passed_first << std::endl;

  // :555-590: names, each database entry once, split by "searched in a previous round"
  Hash<char>* doubled = new Hash<char>;
  doubled->New(16381, 0);
  for (size_t k = 0; k < ids.size(); ++k) {
    char db_name[NAMELEN];
    strcpy(db_name, dbnames[ids[k]]);
    char name[NAMELEN];
    RemoveExtension(name, db_name);
    if (!doubled->Contains(db_name)) {
      doubled->Add(db_name);
      std::pair<int, std::string> result;
      result.first = length[ids[k]];
      result.second = std::string(db_name);
      std::stringstream ss_tmp;
      ss_tmp << name << "__" << 1;
      if (previous_hits->Contains((char*)ss_tmp.str().c_str())) old_prefilter_hits.push_back(result);
      else new_prefilter_hits.push_back(result);
    }
  }
  if ((int)ids.size() >= maxnumdb)
    HH_LOG(WARNING) << "Number of hits passing 2nd prefilter reduced to allowed maximum of " << maxnumdb << ".\n"
                    << "You can increase the allowed maximum using the -maxfilt <max> option.\n";
  delete doubled;
}
