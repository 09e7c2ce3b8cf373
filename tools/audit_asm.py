#!/usr/bin/env python3
"""tools/audit_asm.py -- audit of the inline-asm LDS reads of hhv_stream_kernel in the generated ISA.

hhv_kernels.hip reads the LDS ring through inline asm (LdsColumn) so that hipcc does not put a vmcnt(0) in front of every
ds_read.  hipcc does not count those reads: the destination registers of a read whose wait sits in a LATER statement hold
garbage until that wait, and nothing may touch them in between (a compiler copy or spill there would read stale data).
This script compiles hhv_kernels.hip with --save-temps (or takes an existing .s), and for every instantiation checks:
  1. between an asm block that issues ds_read_* without waiting and the asm block that waits lgkmcnt(0), no instruction
     outside asm blocks reads or writes one of the destination registers;
  2. the kernel's main loop contains no compiler-generated ds_read / flat_load, and - in the variants without global
     loads in the loop (no CELLOFF, MULTI, SS) - no compiler-generated s_waitcnt vmcnt;
  3. no scratch (private segment) is used;
  4. the work queue's ticket (global_atomic_add in an asm block, hhv_stream_kernel.h WorkQueue::draw) is not touched outside
     asm blocks between its draw and the asm vmcnt(0) that precedes its use.
Exit status 0 = clean.  Used by tests/test_asm_audit.py (CPU-only: hipcc cross-compiles without a GPU).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


UNITS = ("hhv_kernels.hip", "hhv_kernels_w32.hip", "hhv_kernels_w16.hip", "hhv_kernels_pair.hip")


def compile_s(workdir):
    """the instantiation units of hhv_stream_kernel (64, 32 and 16 lanes per systolic array) -> one concatenated .s"""
    procs = []
    for u in UNITS:
        src = os.path.join(ROOT, "hh-suite_amd", "csrc", u)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-fvisibility=hidden", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "hh-suite_amd", "csrc"), "--cuda-device-only", "-S", src, "-o",
               os.path.join(workdir, u + ".s")] + os.environ.get("HHV_AUDIT_FLAGS", "").split()   # (measurement builds: -D...)
        procs.append(subprocess.Popen(cmd, cwd=workdir, stderr=subprocess.DEVNULL))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    out = os.path.join(workdir, "hhv_kernels_all.s")
    with open(out, "w") as f:
        for u in UNITS:
            f.write(open(os.path.join(workdir, u + ".s")).read())
            f.write("\n")
    return out


def functions(lines):
    """yield (name, body lines) of every hhv_stream_kernel instantiation"""
    i = 0
    n = len(lines)
    while i < n:
        m = re.match(r"^(_ZN3hhv1[7538]hhv_(?:stream|pair|ss|ss_pair)_kernel\w+):", lines[i])
        if m:
            j = i
            while j < n and not lines[j].strip().startswith("s_endpgm"):
                j += 1
            yield m.group(1), lines[i:j + 1]
            i = j
        i += 1


def audit_function(name, body):
    problems = []
    m = re.search(r"hhv_stream_kernelILi\dELb\dELb\dELb(\d)ELb(\d)ELb(\d)E", name)
    if m and m.group(3) == "1" and "ELi64E" in name:
        return ["%s: a 64-lane secondary-structure variant of hhv_stream_kernel (they run as hhv_ss_kernel)" % name]
    if m is None:   # hhv_ss_kernel<R, LOCAL, BT, CELLOFF, MULTI, FIRSTP>: the table values come out of LDS through inline asm - only the
        m = re.search(r"hhv_ss_kernelILi\dELb\dELb\dELb(\d)ELb(\d)E", name)   # cell-off masks and the carry rows are global loads
    loads_in_loop = m is None or "1" in m.groups()   # (pair kernels: m is None - their waits are placed by hand too, but they are MULTI bodies)
    pending = []         # destination registers of issued, not yet waited-for asm reads, one set per read in issue order
    ticket = []          # destination of the work queue's atomic, not yet waited for
    in_asm = False
    asm_lines = []
    in_loop = False
    for ln, raw in enumerate(body):
        line = raw.strip()
        if "Loop Header" in raw:
            in_loop = True
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            asm_lines = []
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            # LDS returns a wave's reads in order: lgkmcnt(N) leaves at most the N youngest in flight
            for a in asm_lines:
                if a.startswith("ds_read") or a.startswith("ds_bpermute"):
                    pending.append(regs_of(a.split(",")[0]))
                # the work queue's ticket (WorkQueue::draw): in flight on vmcnt until an asm vmcnt(0)
                if a.startswith("global_atomic_add"):
                    ticket.append(regs_of(a.split(",")[0]))
                if re.search(r"vmcnt\(0\)", a):
                    ticket = []
                w = re.search(r"lgkmcnt\((\d+)\)", a)
                if w:
                    n_left = int(w.group(1))
                    pending = pending[len(pending) - n_left:] if n_left else []
            continue
        if in_asm:
            asm_lines.append(line)
            continue
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            continue
        code = line.split(";")[0]
        if pending:
            hit = regs_of(code) & set().union(*pending)
            if hit:
                problems.append("%s: line %d touches v%s while its read is in flight: %s" % (name, ln, sorted(hit), code.strip()))
        if ticket:
            hit = regs_of(code) & set().union(*ticket)
            if hit:
                problems.append("%s: line %d touches v%s while the queue ticket is in flight: %s" % (name, ln, sorted(hit), code.strip()))
        if in_loop:
            op = code.split()[0]
            if op.startswith("ds_read") or op.startswith("flat_load"):
                problems.append("%s: compiler-generated %s in the loop (line %d)" % (name, op, ln))
            if op == "s_waitcnt" and "vmcnt" in code and not loads_in_loop:
                problems.append("%s: compiler-generated '%s' in the loop (line %d)" % (name, code.strip(), ln))
    if pending:
        problems.append("%s: reads still in flight at the end of the function: v%s" % (name, sorted(set().union(*pending))))
    return problems


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
        tmp = None
    else:
        tmp = tempfile.TemporaryDirectory()
        path = compile_s(tmp.name)
    lines = open(path).read().split("\n")
    text = "\n".join(lines)
    problems = []
    count = 0
    for name, body in functions(lines):
        count += 1
        problems += audit_function(name, body)
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", text):
        if int(m.group(1)) != 0:
            problems.append("a kernel uses %s bytes of scratch" % m.group(1))
    for m in re.finditer(r"\.vgpr_spill_count:\s*(\d+)", text):
        if int(m.group(1)) != 0:
            problems.append("a kernel spills %s VGPRs" % m.group(1))
    print("audited %d hhv_stream_kernel / hhv_pair_kernel / hhv_ss_kernel instantiations, %d problems" % (count, len(problems)))
    for p in problems[:50]:
        print("  " + p)
    return 1 if problems or count == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
