"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv (any counter set): time-weighted per-launch means.
    python tools/pmc_summary.py <dir> [substring]"""
import collections, csv, glob, json, os, sys

CUS = 256


def main():
    f = max(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if want not in n:
            continue
        per[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            per[n]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            per[n]["launches"] += 1
    out = {}
    for n, d in sorted(per.items(), key=lambda kv: -kv[1]["ns"]):
        o = {"launches": int(d["launches"]), "us_per_launch": round(d["ns"] / d["launches"] / 1e3, 1)}
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
        if cyc:
            o["clock_ghz"] = round(cyc / d["ns"], 3)
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                o["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (CUS * 4 * cyc), 3)
        wc = d.get("SQ_WAVE_CYCLES", 0)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM"):
            if k in d and wc:
                o[k + "/WAVE_CYCLES"] = round(d[k] / wc, 4)
        for k in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCC_REQ_sum", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
                  "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
            if k in d:
                o[k + "_per_launch"] = round(d[k] / d["launches"], 1)
        out[n] = o
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
