"""profiles/pmc_traffic.json from a PMC summary (tests/_pmc_bench.sh -> tests/_pmc_summary.py).
Usage: python tests/_pmc_traffic.py gpurun_out/TAG/summary.json [reads_per_gpu]"""
import json, os, sys
src = sys.argv[1]
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
d = json.load(open(src))
out = {}
for k in ("k_fused_a", "k_fused_b", "k_symbolize", "k_merge", "k_stats"):
    f, w = d[k]["FETCH_SIZE"], d[k]["WRITE_SIZE"]
    out[k] = {"launches": f["launches"], "fetch_kb_per_launch": round(f["total"] / f["launches"], 1),
              "write_kb_per_launch": round(w["total"] / w["launches"], 1),
              "traffic_bytes_per_launch": int((2 * f["total"] / f["launches"] + w["total"] / w["launches"]) * 1024)}
sym = out["k_symbolize"]
steps = sym["launches"]
valu = sum(d[k]["SQ_INSTS_VALU"]["total"] for k in ("k_fused_a", "k_fused_b") if "SQ_INSTS_VALU" in d[k]) / max(1, steps)
meta = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1",
        "reads_per_gpu": reads,
        "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts half of the bytes of wide streaming reads "
                      "(MI355X_MICROARCH.md, HBM); verified in this very run on k_symbolize, which reads and writes 96.0 MB per launch "
                      "(FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB). For the narrower loads of the fused kernels the factor 2 is an upper bound."
                      % (sym["fetch_kb_per_launch"], sym["write_kb_per_launch"]),
        "valu_wave_insts_per_step": int(valu),
        "kernels": out}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(meta, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
