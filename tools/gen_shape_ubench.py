#!/usr/bin/env python3
"""tools/gen_shape_ubench.py -- writes tools/shape_ubench.hip: one v_add_f32 stream under different kernel shapes.

Workgroup size (64 / 256 threads) x VGPR allocation (32 ... 256, forced with asm clobbers), each run with 1 ... 4 waves per
SIMD as far as the occupancy query admits.  What it showed on MI355X (profiles/r2_shape_ubench.txt): a lone wave issues
one VALU instruction per ~4.5 clk, two waves per SIMD together one per ~2.25 clk whatever their register allocation and
workgroup size, three waves are worse than two (2.9-3.3), four equal two; a kernel that clobbers all 256 VGPRs gets ONE
wave per SIMD.  Measurement aid, not product."""
import os
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# matrix: the same v_add stream under different kernel shapes: workgroup size x VGPR allocation
src=open('tools/shape_ubench.hip','w')
src.write('#include <hip/hip_runtime.h>\n#include <stdio.h>\n')
def stream(kind, nv=32):
    out=[]
    for k in range(256):
        r=10+k%16
        if kind=="mixhi": r=nv-20+k%16   # the same mix on the highest registers of the allocation
        if kind in ("mix","mixhi") and k%5==0: out.append("v_max_f32_e32 v%d, v%d, v1"%(r,r))     # every fifth instruction is a "slow class" one
        else: out.append("v_add_f32_e32 v%d, v%d, v1"%(r,r))
    return "\\n".join(out)+"\\n"
cases=[]
for kind in ("add","mix","mixhi"):
  for wg in (64,128,256):
    for nv in (32,128,248):
        seq=stream(kind,nv)
        if kind=="mixhi" and nv==32: continue
        if wg*nv//64 > 512*4//2 and False: continue
        name='k_%s_wg%d_v%d'%(kind,wg,nv)
        clob=",".join('"v%d"'%k for k in range(0,nv))
        src.write('''__global__ void __launch_bounds__(%d) %s(float* out, int iters) {
  for (int it = 0; it < iters; ++it) { asm volatile("%s" ::: %s); }
  out[blockIdx.x*%d+threadIdx.x] = iters;
}
'''%(wg,name,seq,clob,wg))
        cases.append((name,wg,nv))
src.write('struct C{const char* n; void(*f)(float*,int); int wg; int nv;}; static C cs[]={'+",".join('{"%s",%s,%d,%d}'%(n,n,wg,nv) for n,wg,nv in cases)+'};\n')
src.write(r'''
int main(){ hipDeviceProp_t p; (void)hipGetDeviceProperties(&p,0); int cus=p.multiProcessorCount; float* out; (void)hipMalloc(&out,(size_t)cus*64*256*4);
 hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
 printf("%-16s occ  : clk per wave-instruction per SIMD at waves/SIMD = 1 2 3 4 (when the kernel admits them)\n","case");
 for(auto&c:cs){ int nb=0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb,c.f,c.wg,0); int wpb=c.wg/64; int maxw=nb*wpb/4; printf("%-16s %3d :",c.n,maxw);
  for(int w=1;w<=4;++w){ if(w>maxw){printf("      -");continue;} int blocks=cus*4*w/wpb; int iters=3000;
   hipLaunchKernelGGL(c.f,dim3(blocks),dim3(c.wg),0,0,out,10); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
   hipLaunchKernelGGL(c.f,dim3(blocks),dim3(c.wg),0,0,out,iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1);
   double inst=(double)w*iters*256; printf(" %6.2f", ms*1e-3*p.clockRate*1e3/inst); } printf("\n"); }
 return 0; }
''')
src.close()
