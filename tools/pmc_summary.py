"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean of each counter (+ derived MFMA-busy)."""
import collections, csv, glob, sys
for path in sys.argv[1:]:
    for f in glob.glob(path):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            n = r["Kernel_Name"]
            if "conv" not in n and "wgrad" not in n:
                continue
            n = n.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[n]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for n, d in agg.items():
            m = {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in d.items()}
            s = "  ".join(f"{k}={v:.4g}" for k, v in sorted(m.items()))
            extra = ""
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
                cyc = m["GRBM_GUI_ACTIVE"] / 8
                extra = f"  | clk={cyc / m['_ns']:.2f} GHz  mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}"
            print(f"{f.split('/')[-3]}: {n}: {s}{extra}")
