#!/usr/bin/env python3
"""tools/gen_step_replay_ubench.py -- writes tools/step_replay_ubench.hip: ONE WHOLE STEP of the stream kernel, replayed.

The first unrolled step of hhv_stream_kernel<5, global, score only, 64> is cut from the ISA hipcc generates (common path: the
header block is dropped, its skip branch kept) and wrapped into a loop with the LDS reads pointed at a dummy 16 KiB buffer.  The
variants remove one ingredient each - the branches, the EXEC manipulation, the LDS reads, the s_nop, the DPP modifiers - so that
the cycles per instruction of a lone wave (and of two waves per SIMD) show which ingredient costs what the plain VALU streams of
tools/gen_mix_ubench.py do not explain.  Measurement aid, not product."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
asm = os.path.join(tmp, "k64.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                       "-fvisibility=hidden", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                       "-I" + os.path.join(ROOT, "hh-suite_amd", "csrc"), "--cuda-device-only", "-S",
                       os.path.join(ROOT, "hh-suite_amd", "csrc", "hhv_kernels.hip"), "-o", asm], stderr=subprocess.DEVNULL)
text = open(asm).read().split('\n')
# usage: gen_step_replay_ubench.py [bt]   (default: the score-only kernel; bt: the backtrace variant, VALU-only variants)
BT = len(sys.argv) > 1 and sys.argv[1] == "bt"
name = "_ZN3hhv17hhv_stream_kernelILi5ELb0ELb%dELb0ELb0ELb0ELi64EEEvNS_10StreamArgsE:" % (1 if BT else 0)
b0 = next(i for i, l in enumerate(text) if l.startswith(name))
b1 = next(i for i in range(b0, len(text)) if text[i].startswith(".Lfunc_end"))
L = text[b0:b1]
# prologue: the s_mov of the pinned constants (asm blocks before the loops)
hdr = next(i for i, l in enumerate(L) if "Inner Loop Header" in l)
pro = [l.strip() for l in L[:hdr] if re.match(r"\s*s_mov_b32 s\d+, (0x[0-9a-f]+|\d+)\s*$", l)]
start = hdr - 1  # the label line
labels = [(i, re.match(r"(\.LBB\d+_\d+):", L[i]).group(1)) for i in range(start, len(L)) if re.match(r"\.LBB\d+_\d+:", L[i])]
# step 1 = from the loop header to the head wait behind the join of the "active" block
first_label = labels[0][1]
# blocks in order: 17 (top), 19, [column], 22 (header skip), ..., 30, 31
cbr = [i for i in range(start, len(L)) if "s_cbranch_execz" in L[i]]
active_join = re.search(r"(\.LBB\d+_\d+)", L[cbr[0]]).group(1)           # target of the first execz = join of `if (active)`
hdr_skip_i = next(i for i in cbr if i > cbr[1])                              # execz over the header block
hdr_join = re.search(r"(\.LBB\d+_\d+)", L[hdr_skip_i]).group(1)
i_hdr_join = next(i for i, n in labels if n == hdr_join)
i_active_join = next(i for i, n in labels if n == active_join)
end = next(i for i in range(i_active_join, len(L)) if "s_waitcnt lgkmcnt(0)" in L[i]) + 1
body = L[start:hdr_skip_i + 1] + L[i_hdr_join:end]


def clean(lines, variant):
    out = []
    for l in lines:
        t = l.split(";")[0].rstrip() if not l.strip().startswith(";;") else ""
        t = t.strip()
        if not t or t.startswith(".") and not re.match(r"\.LBB\d+_\d+:", t):
            continue
        m = re.match(r"\.LBB\d+_(\d+):", t)
        if m:
            out.append("%s%%=:" % ("L" + m.group(1) + "_"))
            continue
        t = re.sub(r"\.LBB\d+_(\d+)", lambda m: "L%s_%%=" % m.group(1), t)
        if variant == "nobranch" and t.startswith("s_cbranch"):
            continue
        if variant == "noexec" and re.match(r"s_(and_saveexec|andn2_saveexec|xor_b64|or_b64)", t):
            continue
        if variant == "nolds" and t.startswith("ds_read"):
            continue
        if variant == "nonop" and t.startswith("s_nop"):
            continue
        if variant == "nodpp" and "_dpp" in t:
            t = re.sub(r"v_mov_b32_dpp (v\d+), (v\d+).*", r"v_mov_b32 \1, \2", t)
        if variant == "valuonly" and not t.startswith("v_") and not re.match(r"L\d+_", t):
            continue
        if variant == "valuonly" and t.startswith("v_cmp") and "vcc" in t.split(",")[0]:
            pass
        out.append(t)
    return out


variants = (["valuonly", "v_nocmp", "v_noaddc", "v_max2add", "v_vop3toadd", "v_noslow"] if BT else
            ["nobranch", "nobranch_nodpp", "valuonly", "v_nodpp", "v_noslow"])


def transform(ins, v):
    """variants of the VALU-only stream: which instruction class keeps two waves of a SIMD from overlapping?"""
    out = []
    for t in ins:
        if t.endswith(":") or not t.startswith("v_"):
            out.append(t)
            continue
        op = t.split()[0]
        args = [a.strip() for a in t[len(op):].split(",")]
        slow_all = v in ("v_noslow", "v_noslow_nosgpr", "v_keepdpp")
        if op.startswith("v_addc") and (slow_all or v in ("v_nocmp", "v_noaddc")):
            vs = [a for a in args if re.match(r"v\d+$", a)]
            t = "v_add_u32_e32 %s, %s, %s" % (vs[0], vs[1], vs[2])
        elif op.startswith("v_max3") and (slow_all or v == "v_vop3toadd"):
            t = "v_add_f32_e32 %s, %s, %s" % (args[0], args[1], args[2])
        elif op.startswith("v_max_f32") and (slow_all or v == "v_max2add"):
            t = "v_add_f32_e32 %s, %s, %s" % (args[0], args[1], args[2])
        elif op.startswith("v_cmp") and (slow_all or v == "v_nocmp"):
            srcs = [a for a in args if re.match(r"v\d+$", a)]
            t = "v_add_f32_e32 v247, %s, %s" % (srcs[0] if srcs else "v247", srcs[-1] if srcs else "v247")
        elif op.startswith("v_cndmask") and (slow_all or v == "v_nocmp"):
            srcs = [a for a in args[1:] if re.match(r"v\d+$", a)]
            t = "v_add_f32_e32 %s, %s, %s" % (args[0], srcs[0] if srcs else "v247", srcs[-1] if srcs else "v247")
        elif op in ("v_bfe_u32", "v_and_or_b32") and (slow_all or v == "v_vop3toadd"):
            vs = [a for a in args[1:] if re.match(r"v\d+$", a)]
            t = "v_and_b32_e32 %s, %s, %s" % (args[0], vs[0], vs[0])
        elif op.startswith("v_cvt") and (slow_all or v == "v_nocvt"):
            t = "v_mov_b32_e32 %s, %s" % (args[0], args[1])
        elif "_dpp" in op and ((slow_all and v != "v_keepdpp") or v == "v_nodpp"):
            t = "v_mov_b32_e32 %s, %s" % (args[0], args[1].split()[0])
        if v.startswith("v_bperm") and "_dpp" in op:
            t = "ds_bpermute_b32 %s, v247, %s" % (args[0], args[1].split()[0])
        if v in ("v_nosgpr", "v_noslow_nosgpr") and t.startswith("v_") and not t.startswith("v_cmp") and not t.startswith("v_cndmask"):
            head, rest = t.split(" ", 1)
            rest = re.sub(r"\bs\d+\b", "v247", rest)
            t = head + " " + rest
        if v == "v_onlymuladd" and not (t.startswith("v_mul_f32") or t.startswith("v_add_f32")):
            continue
        out.append(t)
    return out

maxv = 0
for l in body:
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", l):
        maxv = max(maxv, int(m.group(1) or m.group(3)))
src = ['// generated by tools/gen_step_replay_ubench.py - do not edit', '#include <hip/hip_runtime.h>', '#include <stdio.h>', '']
cases = []
for v in variants:
    if v.endswith("_nodpp") and not v.startswith("v_"):
        ins = [re.sub(r"v_mov_b32_dpp (v\d+), (v\d+).*", r"v_mov_b32 \1, \2", t) for t in clean(body, v[:-6])]
    else:
        ins = clean(body, v) if not v.startswith("v_") else transform(clean(body, "valuonly"), v)
    if v.startswith("v_bperm"):
        # the hand-off through the LDS crossbar: where the wait stands, and five selects that put the boundary back into lane 0
        last = max(i for i, t in enumerate(ins) if t.startswith("ds_bpermute"))
        at = last + 1 if v == "v_bperm_wait0" else min(last + 121, len(ins))
        extra = ["s_waitcnt lgkmcnt(0)"]
        if v == "v_bperm_late_fix":
            extra += ["v_cndmask_b32_e64 v%d, v%d, v246, s[90:91]" % (r, r) for r in (240, 241, 242, 243, 244)]
        ins = ins[:at] + extra + ins[at:]
    n = len([x for x in ins if not x.endswith(":")])
    nv = len([x for x in ins if x.startswith("v_")])
    # the whole kernel body is one asm statement that ends the program itself: the replayed code overwrites SGPRs hipcc keeps
    # live (kernarg pointer, ...), so nothing of the C++ epilogue may run after it
    setup = "\\n".join(["s_mov_b32 s100, %0"] + ["s_mov_b32 s%d, 0x7fffffff" % k for k in range(0, 100)] +
                        [x for x in pro if int(re.match(r"s_mov_b32 s(\d+)", x).group(1)) >= 40] +
                        ["v_mov_b32 v%d, 0" % k for k in range(0, 247)] + ["v_mbcnt_lo_u32_b32 v247, -1, 0", "v_mbcnt_hi_u32_b32 v247, -1, v247", "v_add_u32 v247, -1, v247",
                         "v_and_b32 v247, 63, v247", "v_lshlrev_b32 v247, 2, v247", "s_mov_b64 s[90:91], 1"]) + "\\n"
    code = "\\n".join(ins) + "\\n"
    src.append('__global__ void __launch_bounds__(64) k_%s(float* out, int iters) {' % v)
    src.append('  __shared__ float lds[4096]; for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0.f; __syncthreads();')
    src.append('  if (iters < 0) out[threadIdx.x] = lds[threadIdx.x];')
    src.append('  const int delay = 1 + (int)((blockIdx.x * 2654435761u >> 20) % 197u);  // 0 ... ~3000 clk of start offset per wave')
    src.append('  asm volatile("s_mov_b32 s101, %%1\\nDLY%%=:\\ns_nop 7\\ns_nop 7\\ns_sub_u32 s101, s101, 1\\ns_cmp_lg_u32 s101, 0\\ns_cbranch_scc1 DLY%%=\\n%sLOOP%%=:\\ns_mov_b64 exec, -1\\n%ss_sub_u32 s100, s100, 1\\ns_cmp_lg_u32 s100, 0\\ns_cbranch_scc1 LOOP%%=\\ns_endpgm\\n" :: "s"(iters), "s"(delay) : %s);'
               % (setup, code, ",".join(['"v%d"' % k for k in range(0, 252)] + ['"vcc"', '"memory"'])))
    src.append('}')
    cases.append((v, n, nv))
src.append('struct C{const char* n; void(*f)(float*,int); int ni; int nv;}; static C cs[]={' +
           ",".join('{"%s",k_%s,%d,%d}' % (v, v, n, nv) for v, n, nv in cases) + '};')
src.append(r'''
int main(){ hipDeviceProp_t p; (void)hipGetDeviceProperties(&p,0); int cus=p.multiProcessorCount; float* out; (void)hipMalloc(&out,(size_t)cus*16*64*4);
 hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
 printf("%-10s %6s %6s occ : clk per step and clk per instruction at 1 and 2 waves per SIMD (nominal %.0f MHz)\n","variant","instr","valu",p.clockRate/1e3);
 for(auto&c:cs){ int nb=0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb,c.f,64,0); printf("%-10s %6d %6d %3d :",c.n,c.ni,c.nv,nb/4);
  for(int w=1;w<=2;++w){ if(w>nb/4){printf("        -      -");continue;} int blocks=cus*4*w; int iters=3000;
   hipLaunchKernelGGL(c.f,dim3(blocks),dim3(64),0,0,out,10); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
   hipLaunchKernelGGL(c.f,dim3(blocks),dim3(64),0,0,out,iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1);
   double clk=ms*1e-3*p.clockRate*1e3/iters; printf(" %8.0f %6.2f", clk, clk/c.ni); } printf("\n"); }
 return 0; }
''')
open(os.path.join(ROOT, "tools", "step_replay_ubench.hip"), "w").write("\n".join(src))
print("step: %d lines, variants %s, max vgpr %d, pinned %s" % (len(body), cases, maxv, pro))
