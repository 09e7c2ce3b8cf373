// hhv_pack.cpp -- host side of the device data layout: prepared profiles (the arrays the
// reference keeps in HMM::p / HMM::tr, /root/reference src/hhhmm.h:143-150) -> 28-dword column
// records.  Replaces the AoS -> lane-interleaved SoA mapping of HMMSimd::MapHMMVector
// (src/hhhmmsimd.cpp:86-160): the same seven transition slots a DP cell touches
// (src/hhviterbialgorithm.cpp:222-228: slots 2..6 of column j-1, slots 0..1 of column j) are
// stored next to the 20 profile values of column j, so one cell operand set is one 112-byte record.
#include "hhv_pack.h"

#include <string.h>

#include "viterbi_lane.h"

namespace hhv {

// reference enum order of tr[][7], src/hhdecl.h:68
enum { T_M2M = 0, T_M2I = 1, T_M2D = 2, T_I2M = 3, T_I2I = 4, T_D2M = 5, T_D2D = 6 };

bool pack_columns(const float* p, const float* tr, int L, float* out) {
  bool negative = false;
  for (int k = 1; k <= L; ++k) {
    float* w = out + (size_t)(k - 1) * REC_DW;
    const float* src = p + (size_t)k * 20;
    // Profile values are probabilities / odds.  The kernel's log2f4 takes the exponent of a column product with a plain shift
    // (viterbi_lane.h), which is the reference's masked field only for a sign bit of 0: negative values are reported to the
    // caller, and -0.0f is stored as +0.0f (x + 0: every product, sum and log2f4 of the reference comes out the same).
    for (int a = 0; a < 20; ++a) {
      const float v = src[a];
      negative |= v < 0.0f;
      w[a] = v + 0.0f;
    }
    const float* a = tr + (size_t)(k - 1) * 7;
    const float* b = tr + (size_t)k * 7;
    w[REC_M2M] = a[T_M2M];
    w[REC_M2D] = a[T_M2D];
    w[REC_D2M] = a[T_D2M];
    w[REC_D2D] = a[T_D2D];
    w[REC_I2M] = a[T_I2M];
    w[REC_I2I] = b[T_I2I];
    w[REC_M2I] = b[T_M2I];
    w[REC_META] = 0.0f;
  }
  return !negative;
}

static inline void put_i32(float* dst, int32_t v) { memcpy(dst, &v, 4); }

void write_header(float* rec, int32_t index, int32_t L) {
  memset(rec, 0, REC_DW * sizeof(float));
  put_i32(rec + 0, index);
  put_i32(rec + 1, L);
  put_i32(rec + REC_META, META_HDR);
}

bool pack_template(const float* p, const float* tr, int L, int32_t index, float* out, const int8_t* ss_pred,
                   const int8_t* ss_conf, const int8_t* ss_dssp) {
  write_header(out, index, L);
  const bool ok = pack_columns(p, tr, L, out + REC_DW);
  for (int j = 1; j <= L; ++j) {
    int32_t meta = j;
    if (j == L) meta |= META_LAST;
    // src/hhhmmsimd.cpp:132-135: pred_index = (unsigned char) ss_pred * MAXCF + ss_conf, dssp_index = ss_dssp
    const int pred = ss_pred ? (unsigned char)ss_pred[j] : 0, conf = ss_conf ? ss_conf[j] : 0;
    const int dssp = ss_dssp ? (unsigned char)ss_dssp[j] : 0;
    meta |= (int32_t)((unsigned char)(pred * 11 + conf) & META_PRED_MASK) << META_PRED_SHIFT;
    meta |= (int32_t)(dssp & META_DSSP_MASK) << META_DSSP_SHIFT;
    put_i32(out + (size_t)j * REC_DW + REC_META, meta);
  }
  return ok;
}

}  // namespace hhv
