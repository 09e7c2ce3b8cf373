"""sha1 over the kernel sources (hh-suite_amd/csrc/*.hip, *.h, *.cpp), in sorted order: what a committed profile summary was taken
on.  The GPU boxes have no .git, so the summaries carry this instead of a commit id; bench.py recomputes it and says when the
counters it quotes from profiles/ were collected on other sources (`profile_stale`)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha1(only=None):
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "hh-suite_amd", "csrc", "*")))
    for f in files:
        b = os.path.basename(f)
        if not b.endswith((".hip", ".h", ".cpp")):
            continue
        if only and not any(b.startswith(o) for o in only):
            continue
        h.update(b.encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


# the sources each family of kernels is built from
VITERBI = ("hhv_stream_kernel", "viterbi_lane", "hhv_kernels", "hhv_internal")
NEXT_ROWS = ("hhv_prefilter", "hhv_mac", "hhv_internal")
PREP = ("hhv_prep", "hhv_internal", "viterbi_lane")

if __name__ == "__main__":
    print(kernel_sources_sha1())
