#!/usr/bin/env python3
"""tools/audit_mac_asm.py -- ISA audit of the MAC dataflow kernels (hh-suite_amd/csrc/hhv_mac.hip, round 5).

The wavefronts of these kernels wait for each other by polling progress counters in LDS.  Two properties of the generated
code decide whether that is cheap, and neither is visible in the source:
  * the counters must be read with ds_read: through a `volatile` GENERIC pointer the compiler emits flat loads, which are
    counted with the global-memory counter as well, and puts `s_waitcnt vmcnt(0)` into the poll - every poll then waits for
    the wave's outstanding global loads and stores (measured: 3-5 k clocks at every row boundary, NOTES_r5 section 6);
  * no poll loop (the loops around `s_sleep`) may wait for global memory at all.
Checked here for every instantiation of hhv_mac_forward_df_kernel / hhv_mac_backward_df_kernel: no flat_ instruction anywhere,
no vmcnt wait inside a poll loop, no scratch.  CPU-only (hipcc cross-compiles gfx950).  Exit code 0 = clean."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hh-suite_amd", "csrc", "hhv_mac.hip")


def kernels(asm):
    """name -> list of lines of every df kernel body"""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN3hhv2[56]hhv_mac_(?:forward|backward)_df_kernel\S*):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                out[name] = cur
                cur = None
    return out


def poll_loops(lines):
    """(first, last) line indices of the innermost loops that contain s_sleep: from the loop's first label back-referenced by
    a branch after the s_sleep to that branch"""
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        if re.search(r"\bs_sleep\b", l):
            # the closest backward branch after the sleep whose target lies before the sleep
            for j in range(i, min(i + 80, len(lines))):
                m = re.search(r"\bs_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[j]) or re.search(r"\bs_branch\s+(\.LBB\d+_\d+)", lines[j])
                if m and m.group(1) in labels and labels[m.group(1)] <= i:
                    loops.append((labels[m.group(1)], j))
                    break
            else:
                loops.append((max(0, i - 30), min(len(lines) - 1, i + 30)))  # (no branch found: audit a window around the sleep)
    return loops


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else SRC  # (another version of the file: the audit is checked against round 5's first dataflow kernels)
    with tempfile.TemporaryDirectory() as tmp:
        asm_path = os.path.join(tmp, "mac.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
               "-I", os.path.dirname(SRC), "-o", asm_path, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            return 2
        asm = open(asm_path).read()
    ks = kernels(asm)
    bad = 0
    for name, lines in sorted(ks.items()):
        flat = [l.strip() for l in lines if re.search(r"\bflat_(load|store|atomic)", l)]
        scratch = [l.strip() for l in lines if re.search(r"\bscratch_(load|store)|buffer_(load|store)\w* .*offen", l)]
        loops = poll_loops(lines)
        vm = []
        for a, b in loops:
            vm += [lines[i].strip() for i in range(a, b + 1) if re.search(r"s_waitcnt.*vmcnt", lines[i])]
        if not loops:
            print("%s: no poll loop found (s_sleep): the audit does not know this kernel any more" % name)
            bad += 1
        if flat or vm:
            bad += 1
            print("%s: %d flat instruction(s) %s; %d vmcnt wait(s) in poll loops %s" % (name, len(flat), flat[:2], len(vm), vm[:2]))
        if scratch:
            print("%s: note: %d scratch access(es)" % (name, len(scratch)))
    print("audited %d MAC dataflow kernels, %d with findings" % (len(ks), bad))
    return 1 if bad or not ks else 0


if __name__ == "__main__":
    sys.exit(main())
