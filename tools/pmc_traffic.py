"""profiles/pmc_traffic_<profile>.json from a PMC summary (tools/gpu_pmc.sh -> tools/pmc_summary.py).
Usage: python tools/pmc_traffic.py gpurun_out/TAG/summary.json [reads_per_gpu] [profile] [collected]"""
import json, os, sys
src = sys.argv[1]
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10001
profile = sys.argv[3] if len(sys.argv) > 3 else "ecoli30x_simlord_lordec"
collected = sys.argv[4] if len(sys.argv) > 4 else "round 3"
d = json.load(open(src))
out = {}
for k in ("k_poa", "k_gather", "k_fused_a", "k_fused_b", "k_symbolize", "k_trivial", "k_merge", "k_stats", "k_dp2", "k_fuse2"):
    if k not in d or "FETCH_SIZE" not in d[k]:
        continue
    f, w = d[k]["FETCH_SIZE"], d[k]["WRITE_SIZE"]
    out[k] = {"launches": f["launches"], "fetch_kb_per_launch": round(f["total"] / f["launches"], 1),
              "write_kb_per_launch": round(w["total"] / w["launches"], 1),
              "traffic_bytes_per_launch": int((2 * f["total"] / f["launches"] + w["total"] / w["launches"]) * 1024)}
sym = out["k_symbolize"]
steps = sym["launches"]                      # one k_symbolize launch per pass over a batch: the passes of the profiled process
# per step = per pass over a batch, averaged over every pass of the process (the serial pass goes round the rotated batches:
# their launch counts differ by one or two, so totals are divided, not per-launch figures multiplied)
align = [k for k in ("k_poa", "k_fused_a", "k_fused_b") if k in d and "SQ_INSTS_VALU" in d[k]]
sq_steps = d["k_symbolize"]["SQ_INSTS_VALU"]["launches"] if "SQ_INSTS_VALU" in d.get("k_symbolize", {}) else steps
valu = sum(d[k]["SQ_INSTS_VALU"]["total"] for k in align) / max(1, sq_steps)
traffic_step = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in out.values()) / max(1, steps)
sq = {}
if "k_poa" in d:
    for name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
                 "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA"):
        if name in d["k_poa"]:
            sq[name] = d["k_poa"][name]["total"]
meta = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (separate passes) -- python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1 --profile " + profile + " (the serial pass rotates over the three batches of the timed region: per-step figures are averages over them)",
        "collected": collected,
        "profile": profile, "reads_per_gpu": reads,
        "k_poa_sq_counters": sq,
        "k_poa_valu_active_of_wave_cycles": round(sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"], 4) if sq.get("SQ_WAVE_CYCLES") else None,
        "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts half of the bytes of wide streaming reads "
                      "(MI355X_MICROARCH.md, HBM); verified in this very run on k_symbolize, which reads and writes the window bases once "
                      "(FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB per launch). For the narrower loads of the alignment kernels the factor 2 "
                      "is an upper bound; the counters sit at the L2 - fabric boundary, so moves that only travel between L2 and the "
                      "Infinity Cache are counted as well." % (sym["fetch_kb_per_launch"], sym["write_kb_per_launch"]),
        "valu_wave_insts_per_step": int(valu),
        "traffic_bytes_per_step": int(traffic_step),
        "kernels": out}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(meta, open(os.path.join(root, "profiles", "pmc_traffic_%s.json" % profile), "w"), indent=1)
print(json.dumps(meta, indent=1))
