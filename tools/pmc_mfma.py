"""Per-kernel matrix-pipe utilisation from one rocprofv3 --pmc pass over bench.py:

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \\
            --output-format csv -d gpurun_out/pmc_mfma -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/pmc_mfma.py gpurun_out/pmc_mfma > profiles/r01_pmc_mfma_busy.json

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (CUs * 4 SIMDs * shader cycles), shader cycles = GRBM_GUI_ACTIVE / 8 (the counter
is summed over the 8 XCDs); clock_ghz = shader cycles / kernel duration.  Time-weighted means over the launches of a kernel.
"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import pretty

CUS = 256


def main():
    f = max(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(f)):
        n = pretty(r["Kernel_Name"])
        per[n][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"],)
        if key not in seen:
            seen.add(key)
            per[n]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            per[n]["launches"] += 1
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- "
                     "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-phase-b-leg (B=2048 full step, 1x MI355X); summarised by tools/pmc_mfma.py",
           "kernels": {}}
    rows = sorted(per.items(), key=lambda kv: -kv[1]["ns"])
    for n, d in rows:
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or d["ns"] <= 0:
            continue
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        out["kernels"][n] = {"launches": int(d["launches"]), "ms_total": round(d["ns"] / 1e6, 3),
                             "mfma_busy": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (CUS * 4 * cyc), 3),
                             "clock_ghz": round(cyc / d["ns"], 2),
                             "lds_bank_conflict_per_wave_cycle": round(d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, d.get("SQ_WAVE_CYCLES", 1)), 4),
                             "wave_wait_fraction": round(d.get("SQ_WAIT_INST_ANY", 0) / max(1.0, d.get("SQ_WAVE_CYCLES", 1)), 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
