// hhv_sidecar.h -- binary sidecar of an hhm ffindex database (SURVEY.md 8f N1 in the product path).
//
// A cold `hhsearch` spends most of its wall time in HHEntry::getTemplateHMM -> HMM::Read (src/hhdatabase.cpp:398-460,
// src/hhhmm.cpp:202-694): ~57 KB of text per template, parsed number by number.  The drop-in keeps what that parse produces
// next to the database, in "<db>_hhm.ffdata.hhvside" (or $HHV_SIDECAR_DIR/<basename>.hhvside): for every template it had to
// read once, one binary record with
//   * the raw columns exactly as hhv_upload_raw_templates wants them (f[(L+2)*20], tr[(L+1)*7] in the order of
//     src/hhdecl.h:68, Neff_M/I/D[(L+1)*3], Neff_HMM) and the secondary-structure records,
//   * what Hit::initHitFromHMM (src/hhhit.cpp:235-318) and HMM::computeScoreSSMode (src/hhhmm.cpp:1967-1973) read from
//     the HMM: names, family strings, the displayed sequences, the n* indices, L, Neff_HMM,
//   * its validity key: length and FNV-1a hash of the entry's text in the hhm ffdata file, par.nseqdis, and the NULL line of the file
//     (HMM::Read overwrites the caller's background with it, src/hhhmm.cpp:536-546).
// The next process that needs the template reads ~40 KB of binary instead of parsing: no HMM::Read, no MapHMMVector.
// The file is an append-only log (one write() per search under flock), indexed in memory by entry name when it is opened
// (the newest record of a name wins; a record whose key does not match the ffindex entry any more is ignored and the
// template is parsed and appended again).  Only HHM TEXT entries are stored (the result of reading an alignment depends on
// search parameters).  HHV_SIDECAR=0 disables reading and writing.
#ifndef HHV_SIDECAR_H_
#define HHV_SIDECAR_H_

#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace hhv_dropin {

struct SidecarRecord {
  // validity key
  uint64_t ff_hash, ff_length;  // hash of the entry's text in the hhm ffdata file, its length
  int32_t nseqdis;
  float null_pb[20];
  // HMM fields
  int32_t L, n_seqs, n_display, ncons, nfirst, nss_dssp, nsa_dssp, nss_pred, nss_conf;
  float neff_hmm;
  std::string entry_name, name, longname, fam, sfam, fold, cl, file;
  std::vector<std::string> sname, seq;
  // raw columns
  std::vector<float> f, tr, neff;
  std::vector<int8_t> pred, conf, dssp;  // [L+1] or empty
};

class Sidecar {
 public:
  static const uint32_t kRecMagic = 0x31434552u;  // "REC1"

  explicit Sidecar(const std::string& path) : path_(path), map_(NULL), map_bytes_(0), usable_(true), checked_end_(0) { load(); }
  ~Sidecar() {
    if (map_) munmap(map_, map_bytes_);
  }

  // the record of `name` if there is one with this key; false = parse the text
  bool find(const char* name, uint64_t ff_hash, uint64_t ff_length, int nseqdis, const float* pb, SidecarRecord* out) {
    size_t at;
    {  // the index is only written by load(); the lock covers the lookup, the record is decoded outside of it (all
       // threads of a search come through here)
      std::lock_guard<std::mutex> lock(mu_);
      std::unordered_map<std::string, size_t>::const_iterator it = index_.find(name);
      if (it == index_.end() || !map_) return false;
      at = it->second;
    }
    SidecarRecord r;
    if (!decode((const char*)map_ + at, map_bytes_ - at, &r)) return false;
    if (r.ff_hash != ff_hash || r.ff_length != ff_length || r.nseqdis != nseqdis || memcmp(r.null_pb, pb, sizeof(r.null_pb)) != 0)
      return false;
    *out = r;
    return true;
  }

  // queue a record; flush() appends everything queued with one write
  void add(const SidecarRecord& r) {
    std::string blob;
    encode(r, &blob);
    std::lock_guard<std::mutex> lock(mu_);
    pending_.append(blob);
  }
  size_t flush() {
    std::lock_guard<std::mutex> lock(mu_);
    if (pending_.empty() || !usable_) {
      pending_.clear();
      return 0;
    }
    const int fd = open(path_.c_str(), O_RDWR | O_CREAT | O_APPEND, 0644);  // (read: the tail check below)
    if (fd < 0) {  // read-only database directory: the sidecar is an optimisation, not a requirement
      usable_ = false;
      pending_.clear();
      return 0;
    }
    size_t written = 0;
    if (flock(fd, LOCK_EX) == 0) {
      // Under the lock: find the end of the last VALID record.  A file of another format version starts over; a torn tail (a
      // writer killed in the middle of its write, a full disk) is cut off - appended behind it, every later record would be
      // unreachable for load(), which stops at the first invalid one, and the file would grow with every search.
      // (the walk starts where load() or the previous flush() ended - records are only ever appended whole under this
      // lock, so what was valid stays valid; a large sidecar is not re-walked header by header on every search)
      off_t end = valid_end(fd, checked_end_);
      if (end < (off_t)kHeadBytes) {
        if (ftruncate(fd, 0) != 0 || !write_all(fd, file_head(), kHeadBytes)) usable_ = false;
        end = kHeadBytes;
      } else {
        struct stat st;
        if (fstat(fd, &st) == 0 && st.st_size > end && ftruncate(fd, end) != 0) usable_ = false;
      }
      if (usable_) {
        if (write_all(fd, pending_.data(), pending_.size())) written = pending_.size();
        else if (ftruncate(fd, end) != 0) usable_ = false;   // a short write: leave no torn record behind
        checked_end_ = end + (off_t)written;
      }
      flock(fd, LOCK_UN);
    }
    close(fd);
    pending_.clear();
    return written;
  }
  size_t size() const { return index_.size(); }

 private:
  // format version 2: records of version 1 could hold templates truncated by a small -maxres
  enum { kHeadBytes = 16 };
  static const char* file_head() {
    static const char h[kHeadBytes] = {'H', 'H', 'V', 'S', 'I', 'D', 'E', '2', 2, 0, 0, 0, 0, 0, 0, 0};
    return h;
  }

  static bool write_all(int fd, const char* p, size_t n) {
    while (n) {
      const ssize_t w = write(fd, p, n);
      if (w < 0 && errno == EINTR) continue;
      if (w <= 0) return false;
      p += w;
      n -= (size_t)w;
    }
    return true;
  }
  // offset behind the last valid record (record headers only); 0 = no valid file header.  `from` = an offset known to be
  // the end of a valid record of THIS file (0 = none); a file that is shorter than that was started over: walked from its head
  static off_t valid_end(int fd, off_t from) {
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < (off_t)kHeadBytes) return 0;
    char head[kHeadBytes];
    if (pread(fd, head, kHeadBytes, 0) != (ssize_t)kHeadBytes || memcmp(head, file_head(), 8) != 0) return 0;
    off_t at = (from >= (off_t)kHeadBytes && from <= st.st_size) ? from : (off_t)kHeadBytes;
    while (at + 12 <= st.st_size) {
      uint32_t h[3];
      if (pread(fd, h, 12, at) != 12) break;
      if (h[0] != kRecMagic || h[1] < 12 || at + (off_t)h[1] > st.st_size || 12 + (size_t)h[2] > h[1]) break;
      at += h[1];
    }
    return at;
  }

  std::string path_;
  void* map_;
  size_t map_bytes_;
  bool usable_;
  off_t checked_end_;  // end of the last record known to be valid (load(), flush())
  std::mutex mu_;
  std::unordered_map<std::string, size_t> index_;  // entry name -> offset of its newest record
  std::string pending_;

  static void put32(std::string* b, uint32_t v) { b->append((const char*)&v, 4); }
  static void put64(std::string* b, uint64_t v) { b->append((const char*)&v, 8); }
  static void puts_(std::string* b, const std::string& s) {
    put32(b, (uint32_t)s.size());
    b->append(s);
  }
  static void pad8(std::string* b) {
    while (b->size() % 8) b->push_back('\0');
  }
  template <typename T>
  static void putv(std::string* b, const std::vector<T>& v) {
    put32(b, (uint32_t)v.size());
    pad8(b);
    b->append((const char*)v.data(), v.size() * sizeof(T));
    pad8(b);
  }

  static void encode(const SidecarRecord& r, std::string* b) {
    b->clear();
    put32(b, kRecMagic);
    put32(b, 0);  // total bytes, patched below
    puts_(b, r.entry_name);  // first: the index scan reads only this far
    pad8(b);
    put64(b, r.ff_hash);
    put64(b, r.ff_length);
    put32(b, (uint32_t)r.nseqdis);
    b->append((const char*)r.null_pb, sizeof(r.null_pb));
    const int32_t ints[10] = {r.L, r.n_seqs, r.n_display, r.ncons, r.nfirst, r.nss_dssp, r.nsa_dssp, r.nss_pred, r.nss_conf, 0};
    b->append((const char*)ints, sizeof(ints));
    b->append((const char*)&r.neff_hmm, 4);
    puts_(b, r.name);
    puts_(b, r.longname);
    puts_(b, r.fam);
    puts_(b, r.sfam);
    puts_(b, r.fold);
    puts_(b, r.cl);
    puts_(b, r.file);
    for (int k = 0; k < r.n_seqs; ++k) {
      puts_(b, r.sname[k]);
      puts_(b, r.seq[k]);
    }
    pad8(b);
    putv(b, r.f);
    putv(b, r.tr);
    putv(b, r.neff);
    putv(b, r.pred);
    putv(b, r.conf);
    putv(b, r.dssp);
    const uint32_t total = (uint32_t)b->size();
    memcpy(&(*b)[4], &total, 4);
  }

  struct Reader {
    const char* p;
    size_t left;
    bool ok;
    Reader(const char* p, size_t n) : p(p), left(n), ok(true) {}
    bool take(void* dst, size_t n) {
      if (!ok || n > left) return ok = false;
      memcpy(dst, p, n);
      p += n;
      left -= n;
      return true;
    }
    uint32_t u32() {
      uint32_t v = 0;
      take(&v, 4);
      return v;
    }
    uint64_t u64() {
      uint64_t v = 0;
      take(&v, 8);
      return v;
    }
    void str(std::string* s) {
      const uint32_t n = u32();
      if (!ok || n > left) {
        ok = false;
        return;
      }
      s->assign(p, n);
      p += n;
      left -= n;
    }
    void align8(const char* base) {
      const size_t used = (size_t)(p - base), pad = (8 - used % 8) % 8;
      if (pad > left) {
        ok = false;
        return;
      }
      p += pad;
      left -= pad;
    }
    template <typename T>
    void vec(std::vector<T>* v, const char* base) {
      const uint32_t n = u32();
      align8(base);
      if (!ok || (size_t)n * sizeof(T) > left) {
        ok = false;
        return;
      }
      v->assign((const T*)p, (const T*)p + n);
      p += (size_t)n * sizeof(T);
      left -= (size_t)n * sizeof(T);
      align8(base);
    }
  };

  static bool decode(const char* base, size_t avail, SidecarRecord* r) {
    if (avail < 8) return false;
    uint32_t magic, total;
    memcpy(&magic, base, 4);
    memcpy(&total, base + 4, 4);
    if (magic != kRecMagic || total < 8 || total > avail) return false;
    Reader rd(base + 8, total - 8);
    rd.str(&r->entry_name);
    rd.align8(base);
    r->ff_hash = rd.u64();
    r->ff_length = rd.u64();
    r->nseqdis = (int32_t)rd.u32();
    rd.take(r->null_pb, sizeof(r->null_pb));
    int32_t ints[10] = {0};
    rd.take(ints, sizeof(ints));
    r->L = ints[0];
    r->n_seqs = ints[1];
    r->n_display = ints[2];
    r->ncons = ints[3];
    r->nfirst = ints[4];
    r->nss_dssp = ints[5];
    r->nsa_dssp = ints[6];
    r->nss_pred = ints[7];
    r->nss_conf = ints[8];
    rd.take(&r->neff_hmm, 4);
    rd.str(&r->name);
    rd.str(&r->longname);
    rd.str(&r->fam);
    rd.str(&r->sfam);
    rd.str(&r->fold);
    rd.str(&r->cl);
    rd.str(&r->file);
    if (!rd.ok || r->n_seqs < 0 || r->n_seqs > 100000 || r->n_display < 0 || r->n_display > r->n_seqs) return false;
    r->sname.resize(r->n_seqs);
    r->seq.resize(r->n_seqs);
    for (int k = 0; k < r->n_seqs && rd.ok; ++k) {
      rd.str(&r->sname[k]);
      rd.str(&r->seq[k]);
    }
    rd.align8(base);
    rd.vec(&r->f, base);
    rd.vec(&r->tr, base);
    rd.vec(&r->neff, base);
    rd.vec(&r->pred, base);
    rd.vec(&r->conf, base);
    rd.vec(&r->dssp, base);
    if (!rd.ok || r->L < 1) return false;
    const size_t L = (size_t)r->L;
    return r->f.size() == (L + 2) * 20 && r->tr.size() == (L + 1) * 7 && r->neff.size() == (L + 1) * 3 &&
           (r->pred.empty() || r->pred.size() == L + 1) && (r->conf.empty() || r->conf.size() == L + 1) &&
           (r->dssp.empty() || r->dssp.size() == L + 1);
  }

  void load() {
    const int fd = open(path_.c_str(), O_RDONLY);
    if (fd < 0) return;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 16) {
      close(fd);
      return;
    }
    void* m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return;
    if (memcmp(m, file_head(), 8) != 0) {  // another format version: ignored here, started over by the next flush()
      munmap(m, (size_t)st.st_size);
      return;
    }
    map_ = m;
    map_bytes_ = (size_t)st.st_size;
    // index scan: record header + entry name only
    size_t at = kHeadBytes;
    const char* base = (const char*)map_;
    while (at + 12 <= map_bytes_) {
      uint32_t magic, total, nlen;
      memcpy(&magic, base + at, 4);
      memcpy(&total, base + at + 4, 4);
      memcpy(&nlen, base + at + 8, 4);
      if (magic != kRecMagic || total < 12 || at + total > map_bytes_ || 12 + (size_t)nlen > total) break;  // a torn tail
      index_[std::string(base + at + 12, nlen)] = at;
      at += total;
    }
    checked_end_ = (off_t)at;
  }
};

// content hash of an entry's text, eight bytes per step (an entry is ~57 KB; byte-wise FNV would cost as much as the
// binary read it guards)
inline uint64_t sidecar_text_hash(const char* text, size_t n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
  size_t k = 0;
  for (; k + 8 <= n; k += 8) {
    uint64_t w;
    memcpy(&w, text + k, 8);
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 29;
  }
  uint64_t tail = 0;
  if (k < n) memcpy(&tail, text + k, n - k);
  h = (h ^ tail) * 0xC4CEB9FE1A85EC53ull;
  return h ^ (h >> 32);
}

// where the sidecar of an hhm ffdata file lives
inline std::string sidecar_path(const char* hhm_data_filename) {
  const char* dir = getenv("HHV_SIDECAR_DIR");
  std::string data(hhm_data_filename ? hhm_data_filename : "");
  if (dir && *dir) {
    const size_t slash = data.find_last_of('/');
    return std::string(dir) + "/" + (slash == std::string::npos ? data : data.substr(slash + 1)) + ".hhvside";
  }
  return data + ".hhvside";
}
inline bool sidecar_enabled() {
  const char* e = getenv("HHV_SIDECAR");
  return !(e && atoi(e) == 0);
}

}  // namespace hhv_dropin

#endif  // HHV_SIDECAR_H_
