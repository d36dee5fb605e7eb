#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc --save-temps, *.s) for what ROCm 7.2's compiler fault with spilled registers needs.

1. Everywhere: VGPR spill stores that sit in front of the `s_or_b64 exec, exec, ...` that ends a divergent region of the
   same block: such a store runs under the region's partial (or empty) EXEC mask and the lanes switched off there read
   garbage back on the reload.  (Seen in k_poa<16, 6, false>: the window descriptors of half the windows came back from
   scratch as zeros; round 5 met another placement this pattern does not describe -- windows differing from call to call in
   a build whose k_poa instances spilled up to 54 registers.)
2. In k_poa, therefore: ANY spilled VGPR.  The kernel parks what it does not need across its loops in LDS itself
   (park_win, poa_pack.hip) and every instance is built to spill nothing; a change that brings spills back fails the build.
Prints the kernels and lines; exit code 1 if any.

    python -m elector_amd.check_spills file.s [...]
"""
import re
import sys


NO_SPILL_KERNELS = ("k_poaI",)          # (mangled-name fragments) kernels that must not spill a VGPR at all


def check(path):
    bad = []
    kernel = None
    lines = open(path).read().split("\n")
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        if "Folded Spill" in ln and "scratch_store" in ln and kernel and any(k in kernel for k in NO_SPILL_KERNELS):
            bad.append((kernel, i + 1, ln.strip()))
            continue
        if "Folded Spill" in ln and "scratch_store" in ln:
            # forward within the basic block: an exec restore after the store = the store ran under the narrowed mask
            for j in range(i + 1, min(i + 40, len(lines))):
                t = lines[j].strip()
                if re.match(r"^\.LBB\w+:", t) or t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    break
                if re.match(r"^s_or_b64 exec, exec, ", t):
                    bad.append((kernel, i + 1, ln.strip()))
                    break
                if re.match(r"^s_\w+ exec(_lo|_hi)?, ", t) or re.match(r"^s_\w+saveexec\w* ", t):
                    break                                   # a region that opens behind the store: the store ran in front of it
    return bad


def main():
    n = 0
    for p in sys.argv[1:]:
        for kernel, line, text in check(p):
            print("%s:%d: %s: %s" % (p, line, kernel, text))
            n += 1
    print("%d spill store(s) under a narrowed EXEC mask or in a kernel that must not spill" % n)
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())
