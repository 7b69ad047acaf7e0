#!/usr/bin/env python
"""Static footprint of the hot kernels, without a GPU: hipcc -S of one translation unit, then per kernel the instruction count, the
scalar-spill traffic (v_writelane / v_readlane), MFMA count, registers and private-segment size.  With one wave per SIMD a kernel's
lifetime is ~4 cycles x the instructions a wave issues, so these counts are the first thing to look at after a
kernel edit; round 2 found three regressions this way (a 68-byte private segment left by a shared lambda, 395 spill instructions caused
by the trace pointer, ~90 branchy instructions per element in a predicated staging loop).

    python tools/kernel_static.py [ilsx_core.hip] [name-filter ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else "ilsx_core.hip"
    filters = [a for a in sys.argv[1:] if not a.endswith(".hip")] or ["k_mlp2_fwd_split<256, 0, 4", "k_mlp2_bwd_split<256, 0, 4", "k_mlp_bwd_dw", "k_sac_phase_a<256, 0, 4", "k_sac_phase_c<256, 0, 4"]
    out = os.path.join(tempfile.mkdtemp(prefix="kstat_"), "dev.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only",
                           "-S", os.path.join(ROOT, "ilswiss_amd", "csrc", src), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
    demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()   # noqa: E731
    starts = [(i, m.group(1)) for i, l in enumerate(text) for m in [re.match(r"^(_Z\w+):\s", l)] if m]
    print(f"{'kernel':78s} {'instr':>6s} {'mfma':>5s} {'wlane':>6s} {'rlane':>6s} {'vgpr':>5s} {'agpr':>5s} {'priv':>5s}")
    for i, name in starts:
        dn = demangle(name)
        if not any(f in dn for f in filters):
            continue
        end = next(j for j in range(i, len(text)) if text[j].startswith(".Lfunc_end"))
        body = [l for l in text[i + 1:end] if l.strip() and not l.lstrip().startswith((";", "."))]
        meta = "\n".join(text[end:end + 120])
        g = lambda pat: (re.search(pat, meta) or [None, "?"])[1]   # noqa: E731
        vg, ag, pv = g(r"\.num_vgpr, (\d+)"), g(r"\.num_agpr, (\d+)"), g(r"\.private_seg_size, (\d+)")
        cnt = lambda w: sum(w in l for l in body)   # noqa: E731
        print(f"{dn[:78]:78s} {len(body):6d} {cnt('v_mfma'):5d} {cnt('v_writelane'):6d} {cnt('v_readlane'):6d} {vg:>5s} {ag:>5s} {pv:>5s}")


if __name__ == "__main__":
    main()
