"""HBM bytes per launch of the two MFMA kernel classes from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-phase-b-leg
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-phase-b-leg
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 2048 > profiles/r02_pmc_traffic.json

The output carries bench.source_hash() of the kernel sources it was taken on; bench.py refuses a profile of other sources.

Both counters are reported in KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section: 128-B requests of wide
coalesced reads are tallied at 64 B).  A "launch" is one qea_conv_igemm / qea_conv_wgrad call, i.e. the split-K reduction
kernels are charged to the wgrad launch they belong to.
"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import pretty

CLASSES = {
    "conv_igemm": (("conv_igemm_kernel", "conv_igemm_bf3_kernel", "conv_igemm_bf3w_kernel", "conv_igemm_p3_kernel", "conv3x3_halo_kernel",
                    "conv3x3_halo_bf3_kernel", "conv3x3_halo_m16_kernel", "gemm1x1_f16_kernel"), ()),
    "conv_wgrad": (("wgrad_kernel", "wgrad_bf3_kernel", "wgrad_halo_kernel", "wgrad_halo9_bf3_kernel", "wgrad_halo9_spec_kernel", "wgrad_halo9_ring_kernel"), ("splitk_reduce_kernel",)),
}


def totals(d, counter):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for cls, (main, extra) in CLASSES.items():
            if any(k in r["Kernel_Name"] for k in main):
                tot[cls] += float(r["Counter_Value"]) * 1024
                n[cls] += 1
            elif any(k in r["Kernel_Name"] for k in extra):
                tot[cls] += float(r["Counter_Value"]) * 1024
    return tot, n


def _source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.source_hash()


def main():
    fetch, nf = totals(sys.argv[1], "FETCH_SIZE")
    write, nw = totals(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline; 1x MI355X; summarised by tools/pmc_traffic.py",
           "correction": "FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM "
                         "section); WRITE_SIZE as is; both reported in KiB",
           "batch_per_gpu": int(sys.argv[3]) if len(sys.argv) > 3 else 2048, "full_step": True, "source_hash": _source_hash()}
    per_kernel = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for d, counter, col, mul in ((sys.argv[1], "FETCH_SIZE", 0, 2.0), (sys.argv[2], "WRITE_SIZE", 1, 1.0)):
        f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                n = pretty(r["Kernel_Name"])
                per_kernel[n][col] += float(r["Counter_Value"]) * 1024 * mul
                per_kernel[n][2] += col == 0
    out["per_kernel_bytes_per_launch"] = {n: {"launches": v[2], "fetch": round(v[0] / max(1, v[2])), "write": round(v[1] / max(1, v[2]))}
                                          for n, v in sorted(per_kernel.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:24]}
    for cls in CLASSES:
        fb, wb = 2 * fetch[cls] / nf[cls], write[cls] / nw[cls]
        out[cls] = {"launches": nf[cls], "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                    "hbm_bytes_per_launch": round(fb + wb)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
