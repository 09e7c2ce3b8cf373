#!/usr/bin/env python3
"""tools/gen_mix_ubench.py -- writes tools/mix_ubench.hip: issue cost of MIXED VALU sequences on gfx950.

profiles/r2_valu_ubench.txt prices every opcode alone (v_add_f32 2.5 clk, v_max_f32 / v_cmp / v_cndmask / DPP / VOP3 5 clk
per wave-instruction at 2 waves per SIMD).  A sub + alignbit pair measured 5.0 clk for the PAIR, i.e. the full-rate
instruction was free next to the half-rate one.  This generator builds kernels whose loop body is a given pattern of
opcodes (A = v_add_f32, U = v_mul_f32, M = v_max_f32, C = v_cmp+v_addc pair counted as 2, D = DPP mov, S = v_cndmask,
X = v_max3_f32, B = v_sub_f32 + v_alignbit_b32 pair counted as 2) over 16 independent accumulators, so that the cost of a
half-rate instruction can be measured as a function of what stands next to it."""
import sys

PATTERNS = [
    ("A*16", "A" * 16),
    ("M*16", "M" * 16),
    ("MA*8 (alternating)", "MA" * 8),
    ("MMAA*4", "MMAA" * 4),
    ("MMMMAAAA*2", "MMMMAAAA" * 2),
    ("M*8 A*8", "M" * 8 + "A" * 8),
    ("MAAA*4", "MAAA" * 4),
    ("MMMM A*12", "MMMM" + "A" * 12),
    ("MAAAAAAA*2", "MAAAAAAA" * 2),
    ("MM A*14", "MM" + "A" * 14),
    ("MU*8", "MU" * 8),
    ("DA*8", "DA" * 8),
    ("D*8 A*8", "D" * 8 + "A" * 8),
    ("SA*8", "SA" * 8),
    ("S*8 A*8", "S" * 8 + "A" * 8),
    ("XA*8", "XA" * 8),
    ("XAAA*4", "XAAA" * 4),
    ("MD*8", "MD" * 8),
    ("MS*8", "MS" * 8),
    ("C*8 (cmp+addc)", "C" * 8),
    ("CA*5 A", "CA" * 5 + "A"),
    ("B*8 (sub+alignbit)", "B" * 8),
    ("MAM A (2:2 alt) ", "MAMA" * 4),
    ("AAM*5 A", "AAM" * 5 + "A"),
    # phase-sized clusters (the Viterbi column: compare/max chains of phase A and C, then the emission products)
    ("M*128 A*128", "M" * 128 + "A" * 128),
    ("(MA)*128", "MA" * 128),
    ("M*64 A*192", "M" * 64 + "A" * 192),
    ("(MAAA)*64", "MAAA" * 64),
    ("BT-like clustered", ("C" * 7 + "M" * 7 + "A" * 14) * 5 + "A" * 250 + ("C" * 2 + "M" * 2 + "A" * 7) * 5),
    ("BT-like interleaved", ("CMAAAAAAAA") * 45),
    ("BT-like B clustered", ("B" * 7 + "M" * 7 + "A" * 14) * 5 + "A" * 250 + ("B" * 2 + "M" * 2 + "A" * 7) * 5),
    ("BT-like B interleaved", ("BMAAAAAAAA") * 45),
    ("score-like clustered", ("M" * 5 + "A" * 14) * 5 + "A" * 250 + ("M" * 3 + "A" * 7) * 5),
    ("score-like interleaved", ("MAAAAAAAAA") * 40),
    # what a non-VALU instruction costs inside a long VALU stream (per 32 instructions one event)
    ("A*256", "A" * 256),
    ("(A*31 J)*8 taken s_branch", ("A" * 31 + "J") * 8),
    ("(A*31 T)*8 taken scc br", ("A" * 30 + "T") * 8),
    ("(A*31 N)*8 untaken execz", ("A" * 31 + "N") * 8),
    ("(A*31 E)*8 saveexec", ("A" * 31 + "E") * 8),
    ("(A*31 W)*8 waitcnt", ("A" * 31 + "W") * 8),
    ("(A*31 P)*8 s_nop", ("A" * 31 + "P") * 8),
    ("(A*31 L)*8 s_mov", ("A" * 31 + "L") * 8),
    ("(A*15 L)*16 s_mov", ("A" * 15 + "L") * 16),
    ("(A*7 L)*32 s_mov", ("A" * 7 + "L") * 32),
    ("(A*31 D)*8 dpp", ("A" * 31 + "D") * 8),
    ("(A*28 DDDD)*8 dpp", ("A" * 28 + "DDDD") * 8),
    ("(A*27 D*5)*8 dpp", ("A" * 27 + "DDDDD") * 8),
    ("(A*31 d)*8 row_shr", ("A" * 31 + "d") * 8),
    ("(A*27 d*5)*8 row_shr", ("A" * 27 + "ddddd") * 8),
    ("(A*31 b)*8 row_bcast15", ("A" * 31 + "b") * 8),
    ("(A*31 c)*8 row_bcast31", ("A" * 31 + "c") * 8),
    ("(A*31 p)*8 bpermute+wait", ("A" * 31 + "p") * 8),
    ("(A*15 q A*15 w)*8 bperm", ("A" * 15 + "q" + "A" * 15 + "W") * 8),
    ("(A*31 s)*8 permlane32swap", ("A" * 31 + "s") * 8),
    ("(A*31 r)*8 readlane+writelane", ("A" * 30 + "r") * 8),
    ("(A*31 u)*8 quad_perm", ("A" * 31 + "u") * 8),
    ("(A*31 z)*8 ds_swizzle+wait", ("A" * 31 + "z") * 8),
    ("(A*31 m)*8 add_dpp row_shr", ("A" * 31 + "m") * 8),
    ("(A*31 n)*8 add_dpp wave_shr", ("A" * 31 + "n") * 8),
    ("(A*255 D)*2 dpp", ("A" * 255 + "D") * 2),
    ("(A*511 D) dpp", ("A" * 511 + "D")),
    ("A*512", ("A" * 512)),
    ("(A*507 D*5) dpp", ("A" * 507 + "DDDDD")),
    ("(A*31 V)*8 taken vccz", ("A" * 30 + "V") * 8),
    ("(A*127 T)*4 taken scc br", ("A" * 126 + "T") * 4),
    ("(A*127 J)*4 taken s_branch", ("A" * 127 + "J") * 4),
    ("(A*31 Y)*8 ds_write_b128", ("A" * 31 + "Y") * 8),
    ("(A*31 R)*8 ds_read+wait", ("A" * 31 + "R") * 8),
    ("(A*31 G)*8 store8", ("A" * 31 + "G") * 8),
    # dependent chains: the same pattern over fewer independent accumulators (1 = every instruction waits for the one before)
    ("A*256 1 chain", "A" * 256, 1),
    ("A*256 2 chains", "A" * 256, 2),
    ("A*256 4 chains", "A" * 256, 4),
    ("U*256 1 chain", "U" * 256, 1),
    ("M*256 1 chain", "M" * 256, 1),
    ("M*256 2 chains", "M" * 256, 2),
    ("(MAAA)*64 1 chain", "MAAA" * 64, 1),
    ("(MAAA)*64 2 chains", "MAAA" * 64, 2),
    ("(MAAA)*64 4 chains", "MAAA" * 64, 4),
    ("(F)*128 add->max 1 chain", "F" * 128, 1),
    ("(F)*128 add->max 2 chains", "F" * 128, 2),
    ("(F)*128 add->max 5 chains", "F" * 128, 5),
    ("(X)*256 max3 2 chains", "X" * 256, 2),
    ("(A*15 i A*15 W)*8 ds_read", ("A" * 15 + "i" + "A" * 15 + "W") * 8),
    ("(A*63 i A*63 W)*2 ds_read", ("A" * 63 + "i" + "A" * 63 + "W") * 2),
    ("(A*100 D*7 A*340 T) step-like", "A" * 100 + "D" * 7 + "A" * 340 + "T"),
    ("(A*100 D*7 V A*340 T) step-like", "A" * 100 + "D" * 7 + "V" + "A" * 340 + "T"),
    ("(A*440 T) step-like", "A" * 440 + "T"),
]


def emit(pat, nacc=16):
    """nacc = number of independent accumulators the pattern cycles through (1 = one dependent chain)"""
    out = []
    k = 0
    for ch in pat:
        r = "%%%d" % (k % nacc)
        r2 = "%%%d" % ((k + 1) % nacc)
        if ch == "A":
            out.append("v_add_f32 %s, %s, %%16" % (r, r))
        elif ch == "U":
            out.append("v_mul_f32 %s, %s, %%16" % (r, r))
        elif ch == "M":
            out.append("v_max_f32 %s, %s, %%16" % (r, r))
        elif ch == "X":
            out.append("v_max3_f32 %s, %s, %%16, %s" % (r, r, r2))
        elif ch == "D":
            out.append("v_mov_b32_dpp %s, %%16 wave_shr:1 row_mask:0xf bank_mask:0xf" % r)
        elif ch == "S":
            out.append("v_cndmask_b32 %s, %s, %%16, vcc" % (r, r))
        elif ch == "d":
            out.append("v_mov_b32_dpp %s, %%16 row_shr:1 row_mask:0xf bank_mask:0xf" % r)
        elif ch == "b":
            out.append("v_mov_b32_dpp %s, %%16 row_bcast:15 row_mask:0xa bank_mask:0xf" % r)
        elif ch == "c":
            out.append("v_mov_b32_dpp %s, %%16 row_bcast:31 row_mask:0xc bank_mask:0xf" % r)
        elif ch == "u":
            out.append("v_mov_b32_dpp %s, %%16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % r)
        elif ch == "m":
            out.append("v_add_f32_dpp %s, %%16, %s row_shr:1 row_mask:0xf bank_mask:0xf" % (r, r))
        elif ch == "n":
            out.append("v_add_f32_dpp %s, %%16, %s wave_shr:1 row_mask:0xf bank_mask:0xf" % (r, r))
        elif ch == "p":
            out.append("ds_bpermute_b32 %s, %%17, %%16" % r)
            out.append("s_waitcnt lgkmcnt(0)")
            k += 1
            continue
        elif ch == "q":
            out.append("ds_bpermute_b32 v100, %17, %16")
            continue
        elif ch == "z":
            out.append("ds_swizzle_b32 %s, %%16 offset:swizzle(BITMASK_PERM, \\\"00001\\\")" % r)
            out.append("s_waitcnt lgkmcnt(0)")
            k += 1
            continue
        elif ch == "s":
            out.append("v_permlane32_swap_b32 %s, %s" % (r, r2))
        elif ch == "r":
            out.append("v_readlane_b32 s10, %s, 15" % r)
            out.append("v_writelane_b32 %s, s10, 16" % r)
        elif ch == "V":
            out.append("s_mov_b64 vcc, 0")
            out.append("s_cbranch_vccz 1f")
            out.append("1:")
            continue
        elif ch == "Y":
            out.append("ds_write_b128 %17, v[100:103]")
            continue
        elif ch == "J":
            out.append("s_branch 1f")
            out.append("1:")
            continue
        elif ch == "T":
            out.append("s_cmp_eq_u32 0, 0")
            out.append("s_cbranch_scc1 1f")
            out.append("1:")
            continue
        elif ch == "N":
            out.append("s_cbranch_execz 1f")
            out.append("1:")
            continue
        elif ch == "E":
            out.append("s_and_saveexec_b64 s[10:11], exec")
            continue
        elif ch == "W":
            out.append("s_waitcnt lgkmcnt(0)")
            continue
        elif ch == "P":
            out.append("s_nop 0")
            continue
        elif ch == "L":
            out.append("s_mov_b32 s10, s11")
            continue
        elif ch == "R":
            out.append("ds_read_b128 v[100:103], %17")
            out.append("s_waitcnt lgkmcnt(0)")
            continue
        elif ch == "G":
            out.append("global_store_dwordx2 %18, v[100:101], off")
            continue
        elif ch == "i":  # ds_read_b128 issued, waited for by a later W
            out.append("ds_read_b128 v[100:103], %17")
            continue
        elif ch == "F":  # dependent v_add -> v_max pair on one register (phase A / C chains)
            out.append("v_add_f32 %s, %s, %%16" % (r, r))
            out.append("v_max_f32 %s, %s, %%16" % (r, r))
        elif ch == "C":
            out.append("v_cmp_gt_f32_e32 vcc, %s, %%16" % r)
            out.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (r2, r2, r2))
            k += 1
        elif ch == "B":
            out.append("v_sub_f32 %s, %%16, %s" % (r, r2))
            out.append("v_alignbit_b32 %s, %s, %s, 31" % (r2, r2, r))
            k += 1
        k += 1
    return out


def main():
    src = ['// generated by tools/gen_mix_ubench.py - do not edit', '#include <hip/hip_runtime.h>', '#include <stdio.h>', '']
    for idx, item in enumerate(PATTERNS):
        name, pat = item[0], item[1]
        ins = emit(pat, *item[2:])
        body = "\\n".join(ins) + "\\n"
        src.append('__global__ void __launch_bounds__(256) k%d(float* out, int iters, float bb) {' % idx)
        src.append('  __shared__ float lds[1024]; lds[threadIdx.x] = bb; __syncthreads();')
        src.append('  float a[16]; float b = bb; unsigned ldsaddr = (threadIdx.x & 63) * 16; float* gp = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 2;')
        src.append('  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;')
        src.append('  for (int it = 0; it < iters; ++it) {')
        outs = ", ".join('"+v"(a[%d])' % i for i in range(16))
        rep = 4 if len(ins) <= 32 else 1
        src.append('    asm volatile(%s : %s : "v"(b), "v"(ldsaddr), "v"(gp) : "vcc", "s10", "s11", "v100", "v101", "v102", "v103", "memory");' % (" ".join(['"%s"' % body] * rep), outs))
        src.append('  }')
        src.append('  float s = 0; for (int i = 0; i < 16; ++i) s += a[i];')
        src.append('  out[blockIdx.x * blockDim.x + threadIdx.x] = s;')
        src.append('}')
        src.append('')
    src.append('struct Case { const char* name; void (*fn)(float*, int, float); int n_inst; };')
    src.append('static Case cases[] = {')
    for idx, item in enumerate(PATTERNS):
        name, pat = item[0], item[1]
        n = len([x for x in emit(pat, *item[2:]) if not x.endswith(':')])
        src.append('  {"%s", k%d, %d},' % (name, idx, (4 if n <= 32 else 1) * n))
    src.append('};')
    src.append(r'''
int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out;
  (void)hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float) * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("device %s  CUs %d  clock %.0f MHz; clk per wave-instruction per SIMD\n", prop.name, cus, prop.clockRate / 1e3);
  printf("%-24s %8s %8s %8s\n", "pattern", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
  const int iters = 2000;
  for (auto& c : cases) {
    printf("%-24s", c.name);
    for (int wps : {1, 2, 4}) {
      const int blocks = cus * wps;
      hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double inst = (double)blocks * 4 * iters * c.n_inst;  // wave-instructions
      const double clk = prop.clockRate * 1e3;
      printf(" %8.2f", 4.0 / (inst / (ms * 1e-3) / clk / cus));
    }
    printf("\n");
  }
  return 0;
}
''')
    open(sys.argv[1] if len(sys.argv) > 1 else "tools/mix_ubench.hip", "w").write("\n".join(src))


if __name__ == "__main__":
    main()
