"""profiles/pmc_traffic.json from a PMC summary (tests/_r2_pmc.sh TAG traffic -> tests/_pmc_summary.py).
Usage: python tests/_pmc_traffic.py gpurun_out/TAG/summary.json [reads_per_gpu] [profile]"""
import json, os, sys
src = sys.argv[1]
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10001
profile = sys.argv[3] if len(sys.argv) > 3 else "ecoli30x_simlord_lordec"
d = json.load(open(src))
out = {}
for k in ("k_poa", "k_fused_a", "k_fused_b", "k_symbolize", "k_trivial", "k_merge", "k_stats"):
    if k not in d or "FETCH_SIZE" not in d[k]:
        continue
    f, w = d[k]["FETCH_SIZE"], d[k]["WRITE_SIZE"]
    out[k] = {"launches": f["launches"], "fetch_kb_per_launch": round(f["total"] / f["launches"], 1),
              "write_kb_per_launch": round(w["total"] / w["launches"], 1),
              "traffic_bytes_per_launch": int((2 * f["total"] / f["launches"] + w["total"] / w["launches"]) * 1024)}
sym = out["k_symbolize"]
steps = sym["launches"]
align = [k for k in ("k_poa", "k_fused_a", "k_fused_b") if k in d and "SQ_INSTS_VALU" in d[k]]
valu = sum(d[k]["SQ_INSTS_VALU"]["total"] / d[k]["SQ_INSTS_VALU"]["launches"] * (d[k]["SQ_INSTS_VALU"]["launches"] // max(1, steps))
           for k in align)
traffic_step = sum(v["traffic_bytes_per_launch"] * (v["launches"] // max(1, steps)) for v in out.values())
meta = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (separate passes) -- python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1",
        "profile": profile, "reads_per_gpu": reads,
        "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts half of the bytes of wide streaming reads "
                      "(MI355X_MICROARCH.md, HBM); verified in this very run on k_symbolize, which reads and writes the window bases once "
                      "(FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB per launch). For the narrower loads of the alignment kernels the factor 2 "
                      "is an upper bound; the counters sit at the L2 - fabric boundary, so moves that only travel between L2 and the "
                      "Infinity Cache are counted as well." % (sym["fetch_kb_per_launch"], sym["write_kb_per_launch"]),
        "valu_wave_insts_per_step": int(valu),
        "traffic_bytes_per_step": int(traffic_step),
        "kernels": out}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(meta, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(meta, indent=1))
