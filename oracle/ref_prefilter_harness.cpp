// oracle/ref_prefilter_harness.cpp -- TEST INFRASTRUCTURE: the reference's two prefilter kernels
//   Prefilter::ungapped_sse_score   src/hhprefilter.cpp:214-278   (gapless profile/sequence score, uint8 SIMD)
//   Prefilter::swStripedByte        src/hhprefilter.cpp:70-212    (striped Smith-Waterman, uint8 SIMD, Farrar/Zhao)
// called on a plain [220][Lq] byte profile (219 column states + the ANY state): the harness stripes the
// profile exactly like Prefilter::stripe_query_profile does (src/hhprefilter.cpp:386-425) and calls the
// private member functions (they use no member data; -fno-access-control, calloc'ed object shell).
// hhprefilter.cpp includes the generated resource header cs219.lib.h; oracle/Makefile generates it from
// $(REF)/data/cs219.lib (the same bytes the reference's cmake ResourceCompiler embeds).
#include <cstdlib>
#include <cstring>

#include "hhprefilter.h"

namespace {
unsigned char* stripe(const unsigned char* plain, int Lq, int offset, int* W_out) {
  const int element_count = VECSIZE_INT * 4;
  const int W = (Lq + element_count - 1) / element_count;
  unsigned char* qc = (unsigned char*)malloc_simd_int((size_t)220 * (Lq + element_count));
  for (int a = 0; a < 220; ++a) {
    int h = a * W * element_count;
    for (int i = 0; i < W; ++i) {
      int j = i;
      for (int k = 0; k < element_count; ++k) {
        qc[h++] = (j >= Lq) ? (unsigned char)offset : plain[(size_t)a * Lq + j];
        j += W;
      }
    }
  }
  *W_out = W;
  return qc;
}
}  // namespace

extern "C" {

int ref_prefilter_vecbytes() { return VECSIZE_INT * 4; }

// plain[220][Lq] (state-major), seqs: n_db sequences concatenated (values 0..219), offsets[n_db+1]
int ref_prefilter_scores(const unsigned char* plain, int Lq, const unsigned char* seqs, const long* offsets, int n_db,
                         int score_offset, int gap_init, int gap_extend, int* ungapped, int* gapped) {
  int W;
  unsigned char* qc = stripe(plain, Lq, score_offset, &W);
  const int element_count = VECSIZE_INT * 4;
  simd_int* ws = (simd_int*)malloc_simd_int(3 * (Lq + element_count));
  Prefilter* pf = (Prefilter*)calloc(1, sizeof(Prefilter));
  for (int n = 0; n < n_db; ++n) {
    unsigned char* s = const_cast<unsigned char*>(seqs + offsets[n]);
    const int len = (int)(offsets[n + 1] - offsets[n]);
    if (ungapped) ungapped[n] = pf->ungapped_sse_score(qc, Lq, s, len, (unsigned char)score_offset, ws);
    if (gapped) gapped[n] = pf->swStripedByte(qc, Lq, s, len, gap_init, gap_extend, ws, ws + W, ws + 2 * W, score_offset);
  }
  free(pf);
  free(ws);
  free(qc);
  return 0;
}

}  // extern "C"
