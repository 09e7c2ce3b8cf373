#!/usr/bin/env python3
"""tools/gen_replay_ubench.py -- writes tools/replay_ubench.hip: the emission phase of the stream kernel, replayed.

Takes the ISA hipcc generates for hhv_stream_kernel<5, global, score only, 64> (compiled here with --save-temps semantics:
-S), cuts out the instructions behind the wait for the profile reads (phase B of the first unrolled step: five 20-term
products, log2f4) and wraps them - registers, literals and all - into a loop, next to synthetic streams with the same shape
(plain v_add, the product/sum pairs with one or several temporaries, software-pipelined pairs).  Answers: does the real
instruction sequence issue slower than a plain v_add stream (it does not: 4.8 vs 4.6 clk per wave-instruction for a lone
wave, profiles/r2_replay_ubench.txt)?  Measurement aid, not product.  The kernels clobber v0..v255, which leaves hipcc no
register for its own values: they get one wave per SIMD whatever the launch asks for (see tools/gen_shape_ubench.py for
the occupancy matrix)."""
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
name = "_ZN3hhv17hhv_stream_kernelILi5ELb0ELb0ELb0ELb0ELb0ELi64EEEvNS_10StreamArgsE:"
b0 = next(i for i, l in enumerate(text) if l.startswith(name))
b1 = next(i for i in range(b0, len(text)) if text[i].startswith(".Lfunc_end"))
L = text[b0:b1]
# the first wait for the profile reads inside the inner loop = before_B of the first unrolled step
s0 = next(i for i, l in enumerate(L) if "Inner Loop Header" in l)
waits = [i for i in range(s0, len(L)) if "s_waitcnt lgkmcnt(0)" in L[i]]
start = waits[0] + 2
ins=[]
i=start
while len(ins)<300 and i<len(L):
    t=L[i].strip(); i+=1
    if not t or t[0] in ';.': continue
    if t.startswith('s_') or t.startswith('ds_') or 'ASM' in t: continue
    ins.append(t)
def emit(name, seq):
    body="\\n".join(seq)+"\\n"
    clob=",".join('"v%d"'%k for k in range(0,256))
    return '''__global__ void __launch_bounds__(64) %s(float* out, int iters) {
  asm volatile("s_mov_b32 s59, 0x7fffff\\n" ::: "s59");
  for (int it = 0; it < iters; ++it) {
    asm volatile("%s" ::: %s, "s59", "vcc");
  }
  out[blockIdx.x*64+threadIdx.x] = iters;
}
''' % (name, body, clob)
phaseB=ins[:275]
# variant: pure v_add stream with the same operands pattern replaced: v_add_f32 vK, vK, v1 over 16 regs
simple=["v_add_f32_e32 v%d, v%d, v1"%(10+k%16,10+k%16) for k in range(275)]
# variant: phase B with literals replaced by a register (v2)
nolit=[re.sub(r'0x[0-9a-f]+','v2',t) if re.search(r'0x[0-9a-f]{8}',t) else t for t in phaseB]
nolit=[t.replace('v_and_or_b32','v_and_or_b32') for t in nolit]
# variant: only the mul/add pairs (drop log part)
muladd=[t for t in phaseB if (t.startswith('v_mul_f32') or t.startswith('v_add_f32')) and '0x' not in t and '-1.0' not in t]
# variant: same mul/add but temp register rotated over v221..v228 to remove the WAW/RAW distance-1 on a single temp
rot=[]
k=0
cur={}
for t in muladd:
    m=re.match(r'(v_mul_f32_e32) (v\d+), (v\d+), (v\d+)',t)
    rot.append(t)
src=open(os.path.join(ROOT, 'tools', 'replay_ubench.hip'),'w')
src.write('#include <hip/hip_runtime.h>\n#include <stdio.h>\n')
cases=[('phaseB',phaseB),('simple_add',simple),('phaseB_nolit',nolit),('muladd_only',muladd)]
# register-distance variants of a synthetic dot: mul into temp T, add acc += T ; temp fixed vs rotating; operands from banks
def dot(temp_rot, q0, t0, nacc):
    seq=[]
    for k in range(128):
        T=221+(k%temp_rot)
        seq.append("v_mul_f32_e32 v%d, v%d, v%d"%(T, q0+(k%100), t0+(k%20)))
        seq.append("v_add_f32_e32 v%d, v%d, v%d"%(130+(k%nacc), T, 130+(k%nacc)))
    return seq
cases+= [('dot_temp1',dot(1,20,142,4)),('dot_temp4',dot(4,20,142,4)),('dot_temp8',dot(8,20,142,4))]
# mul and add streams not dependent at distance 1: software pipelined (add uses the temp of the previous pair)
def dot_sw(nt):
    seq=[]
    for k in range(128):
        T=221+(k%nt); Tp=221+((k-1)%nt)
        seq.append("v_mul_f32_e32 v%d, v%d, v%d"%(T, 20+(k%100), 142+(k%20)))
        seq.append("v_add_f32_e32 v%d, v%d, v%d"%(130+(k%4), Tp, 130+(k%4)))
    return seq
cases+=[('dot_swpipe2',dot_sw(2)),('dot_swpipe4',dot_sw(4))]
for n,sq in cases: src.write(emit('k_'+n,sq))
src.write('struct C{const char* n; void(*f)(float*,int); int ni;}; static C cs[]={'+",".join('{"%s",k_%s,%d}'%(n,n,len(sq)) for n,sq in cases)+'};\n')
src.write(r'''
int main(){ hipDeviceProp_t p; (void)hipGetDeviceProperties(&p,0); int cus=p.multiProcessorCount; float* out; (void)hipMalloc(&out,(size_t)cus*16*64*4);
 hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
 printf("%-16s %6s %9s %9s   (clk at %.0f MHz per wave-instruction per SIMD)\n","case","instr","1 w/SIMD","2 w/SIMD",p.clockRate/1e3);
 for(auto&c:cs){ printf("%-16s %6d",c.n,c.ni); for(int w: {1,2}){ int blocks=cus*4*w; int iters=4000;
   hipLaunchKernelGGL(c.f,dim3(blocks),dim3(64),0,0,out,10); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
   hipLaunchKernelGGL(c.f,dim3(blocks),dim3(64),0,0,out,iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1);
   double inst=(double)w*iters*c.ni; printf(" %9.2f", ms*1e-3*p.clockRate*1e3/inst); } printf("\n"); }
 return 0; }
''')
src.close()
print(len(phaseB),len(muladd)); print("\n".join(phaseB[:12]))
