"""The stream kernel reads LDS through inline asm that hipcc does not count: audit the generated
ISA of every instantiation (tools/audit_asm.py) - nothing touches a destination register while its read is in flight,
no compiler-generated LDS read or vmcnt wait in the loop of the plain variants, no scratch.  CPU-only: hipcc
cross-compiles gfx950 without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_kernel_isa_audit():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "audited 406" in out.stdout, out.stdout   # W = 64: 60 + 30 (first-strip kernels) hhv_stream_kernel and as many hhv_ss_kernel; 60 each of W = 32 and W = 16; 58 pair kernels (16 x 4 chain positions - 6 that spill) + 48 hhv_ss_pair_kernel (strips of three / four rows)


def test_mac_dataflow_kernels_poll_lds_only():
    """tools/audit_mac_asm.py: the progress counters of the MAC dataflow kernels are read with ds instructions and no poll loop
    waits for global memory (a volatile generic pointer had made every poll a flat load + s_waitcnt vmcnt(0): 3-5 k clocks a row)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_mac_asm.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    # forward: local / global x (plain, ring) = 4; backward: the same x (with / without the -omat lists) = 8
    # (round 6: the instantiations with the template's copy in LDS are gone - the dataflow kernels read it from global memory)
    assert "audited 12 MAC dataflow kernels, 0 with findings" in out.stdout, out.stdout
