// hhv_pack.h -- host packer of the 28-dword column records (see hhv_pack.cpp, viterbi_lane.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace hhv {

// columns 1..L of an HMM given as p[(L+1)*20], tr[(L+1)*7] -> out[L*28] (meta = 0).
// Returns false if a profile value is negative (the caller refuses the profile); -0.0f is stored as +0.0f.
bool pack_columns(const float* p, const float* tr, int L, float* out);
// header record of a template (index, L); index -1 = terminal header
void write_header(float* rec, int32_t index, int32_t L);
// header + L column records with meta (j, LAST flag, secondary-structure indices) -> out[(L+1)*28]
// ss_pred/ss_conf/ss_dssp: [L+1] or null (= the zeros the reference keeps for HMMs without SS records)
bool pack_template(const float* p, const float* tr, int L, int32_t index, float* out, const int8_t* ss_pred = nullptr,
                   const int8_t* ss_conf = nullptr, const int8_t* ss_dssp = nullptr);

}  // namespace hhv
