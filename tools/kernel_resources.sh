#!/bin/bash
# Registers, spills, scratch and LDS of every kernel as the compiler reports them (-Rpass-analysis=kernel-resource-usage),
# one line per kernel -> stdout.  CPU only (cross-compiles for gfx950).  Usage: tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
for f in $R/elector_amd/csrc/*.hip; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -I$R/include -I$R/elector_amd/csrc -c $f -o $T/$(basename $f).o \
      -Rpass-analysis=kernel-resource-usage 2> $T/$(basename $f).log ) &
  while [ $(jobs -r | wc -l) -ge 5 ]; do sleep 1; done
done
wait
echo "# hipcc -Rpass-analysis=kernel-resource-usage over elector_amd/csrc/*.hip (gfx950)"
python3 - $T <<'PY'
import glob, re, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.log")):
    cur = {}
    for ln in open(f, errors="replace"):
        m = re.search(r"remark:\s+(?:Function Name: (\S+)|([A-Za-z \[\]/]+): (\d+))", ln)
        if not m: continue
        if m.group(1):
            if cur: print("\t   ".join("%s: %s" % kv for kv in cur.items()))
            cur = {"Name": m.group(1)}
        elif m.group(2).strip() in ("VGPRs", "Occupancy [waves/SIMD]", "VGPRs Spill", "ScratchSize [bytes/lane]", "SGPRs Spill", "LDS Size [bytes/block]"):
            cur[m.group(2).strip()] = m.group(3)
    if cur: print("\t   ".join("%s: %s" % kv for kv in cur.items()))
PY
rm -rf $T
