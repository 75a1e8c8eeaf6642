"""Rewrites the 'Numbers of the final build' paragraph of DESIGN.md (and the cross-check line of profiles/README.md) from the
committed profiles/r04_* files, so that the documents quote exactly what the files hold."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, *a)
DOM = "conv3x3_halo_m16_kernel<128, false, 0, 0, false>"
WG = "wgrad_halo9_ring_kernel"
sys_path = os.path.join(ROOT, "tools")
import sys
sys.path.insert(0, sys_path)
from kname import pretty
BEGIN, END = "<!-- numbers:begin -->", "<!-- numbers:end -->"


def main():
    d = json.loads(open(P("profiles", "r04_bench_b2048.json")).read().strip().splitlines()[-1])
    u = json.loads(open(P("profiles", "r04_bench_b2048_under_rocprof_single_stream.json")).read().strip().splitlines()[-1])
    tr, total = None, 0.0
    for row in csv.DictReader(open(P("profiles", "r04_kernel_stats_b2048_single_stream.csv"))):
        total += float(row["TotalDurationNs"])
        if pretty(row["Name"]) == DOM:
            tr = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
    t = json.load(open(P("profiles", "r04_pmc_traffic.json")))
    m = json.load(open(P("profiles", "r04_pmc_mfma_busy.json")))["kernels"]
    k0, k1 = m[DOM], m[WG]
    r = d["roofline"]
    lc, wg, nf, bf = r["launch_class"], d["kernels"]["conv_wgrad"], r["native_fp32"], r["split_bf16"]
    kt = t["per_kernel_bytes_per_launch"][DOM]
    kb = kt["fetch"] + kt["write"]
    cb = d["cpu_baseline"]
    nums = f'''{BEGIN}
Numbers of the final build (1× MI355X, `profiles/r04_bench_b2048.json`): **{d['value'] / 1e3:.2f} k patch-images/s for the full minibatch step at
B = 2048** ({d['ms_per_step']:.1f} ms incl. the CER update; {d['overlap']['ms_per_step_single_stream']:.1f} ms single-stream; {d['full_step_without_cer_update']['ms_per_step']:.1f} ms = {d['full_step_without_cer_update']['value'] / 1e3:.2f} k without the CER update, the round-2
timed region; round 3: 104.3 ms, round 2: 150.6 ms), of which Phase B alone is {d['phase_b']['ms_per_step']:.1f} ms = {d['phase_b']['value'] / 1e3:.2f} k img/s ({d['phase_b']['end_to_end_tflops']:.1f} TFLOP/s end to end on the
reference-faithful 9.846 GFLOP/img); configs[1] (B = 512, Phase B) {d['configs1_b512']['value'] / 1e3:.2f} k img/s ({d['configs1_b512']['ms_per_step']:.1f} ms); `--select_before_clean` {d['full_step_select_before_clean']['value'] / 1e3:.2f} k.
The same steps on the same box in the other two arithmetic forms: three-way bf16 split (round 2's) {bf['value'] / 1e3:.2f} k img/s
({bf['ms_per_step_single_stream']:.1f} ms single-stream; its dominant kernel {bf['achieved']:.1f} TFLOP/s = {bf['frac']:.3f} of 419.4), native fp32 MFMA {nf['value'] / 1e3:.2f} k (conv class {nf['conv_igemm_tflops']:.1f} TFLOP/s =
{nf['frac']:.3f} of 157.3).
`roofline` = the dominant kernel `{DOM}` ({100 * r['share_of_step']:.1f} % of the step, {r['launches_per_step']:.0f} launches/step): **{r['achieved']:.1f} TFLOP/s
fp32-equivalent, `frac` {r['frac']:.3f}** of 838.9 (fp16 dense ÷ 3; {r['frac_of_the_six_mfma_peak']:.3f} of the 419.4 the six-MFMA form was priced against); average launch {r['avg_launch_us']:.1f} µs
from bench.py's HIP events — the rocprofv3 trace of a run made of identical full steps gives {tr[1]:.1f} µs for the same kernel
(`r04_kernel_stats_b2048_single_stream.csv`, {tr[0]} calls; the JSON line of that very run: {u['roofline']['avg_launch_us']:.1f}); HBM traffic {kb / 1e9:.2f} GB per launch against
{r['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic ({kb / r['algorithmic_bytes_per_launch']:.2f}×). `roofline.launch_class` = all {lc['launches_per_step']:.0f} `qea_conv_igemm` launches of a step: {lc['achieved']:.1f} TFLOP/s, {lc['frac']:.3f} of the
{lc['peak']:.1f} blend ({100 * lc['split_f16_flop_fraction']:.1f} % of the flops in the fp16 split), {t['conv_igemm']['hbm_bytes_per_launch'] / 1e9:.2f} GB of HBM traffic per launch against {lc['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic
({t['conv_igemm']['hbm_bytes_per_launch'] / lc['algorithmic_bytes_per_launch']:.2f}×). `kernels.conv_wgrad`: {wg['tflops']:.1f} TFLOP/s over {wg['launches_per_step']:.0f} launches ({wg['ms_per_step']:.1f} ms; round 3: 273.6 TFLOP/s, 25.2 ms; round 2: 188.6, 36.5), `frac` {wg['frac']:.3f} of its
{wg['peak']:.1f} blend, HBM {t['conv_wgrad']['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch against {wg['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic ({t['conv_wgrad']['hbm_bytes_per_launch'] / wg['algorithmic_bytes_per_launch']:.2f}×). BiLSTM steps {d['kernels']['lstm_step']['ms_per_step']:.1f} ms ({d['kernels']['lstm_step']['launches_per_step']:.0f} launches).
`cpu_baseline`: {cb['value']:.1f} img/s at B = {cb['batch']} (B = 128: {cb['b128']['value']:.1f}; {cb['cores']} threads). PMC (`r04_pmc_mfma_busy.json`, under the profiler): dominant
kernel {k0['mfma_busy']} MFMA-busy at a held {k0['clock_ghz']} GHz = {k0['mfma_busy'] * k0['clock_ghz'] / 2.4:.3f} of the peak (round 3: 0.647 at 1.81 = 0.488), no LDS bank conflicts (round 3's swizzle family: 0.135 per wave cycle); nine-tap wgrad (producer / consumer form) {k1['mfma_busy']} at {k1['clock_ghz']} GHz (round 3: 0.412 at 2.2); sum of kernel time in the
single-stream trace {total / 13e6:.1f} ms per step.
{END}'''
    # single-stream time by class, from the same trace
    cls = {"conv": 0.0, "wgrad": 0.0, "bn": 0.0, "lstm": 0.0, "pool": 0.0, "c1head": 0.0, "absmax": 0.0, "other": 0.0}
    launches = 0
    for row in csv.DictReader(open(P("profiles", "r04_kernel_stats_b2048_single_stream.csv"))):
        n, t = pretty(row["Name"]), float(row["TotalDurationNs"]) / 13e6
        launches += int(row["Calls"]) / 13
        if ("wgrad" in n and "c1" not in n) or "splitk" in n or "bias_finalize" in n:
            cls["wgrad"] += t
        elif any(k in n for k in ("conv3x3", "conv_igemm", "gemm1x1")):
            cls["conv"] += t
        elif any(k in n for k in ("bn_", "colreduce", "partials", "colsum")):
            cls["bn"] += t
        elif "lstm" in n:
            cls["lstm"] += t
        elif "maxpool" in n:
            cls["pool"] += t
        elif any(k in n for k in ("c1_", "conv_c1", "head_")):
            cls["c1head"] += t
        elif "absmax" in n:
            cls["absmax"] += t
        else:
            cls["other"] += t
    tot = sum(cls.values())
    hbm = cls["bn"] + cls["pool"] + cls["c1head"] + cls["absmax"]
    classes = f"""<!-- classes:begin -->
Single-stream time by class (`r04_kernel_stats_b2048_single_stream.csv`, {tot:.1f} ms of kernel time per step in {launches:.0f} launches; round 3:
108.0 ms in 1 042): conv + GEMM {100 * cls['conv'] / tot:.0f} % ({cls['conv']:.1f} ms; round 3: 53.7, round 2: 83), wgrad incl. slab reductions {100 * cls['wgrad'] / tot:.0f} % ({cls['wgrad']:.1f}; 26.5; 36),
BatchNorm (apply incl. the fused pools, backward incl. the fused pool backward, column sums) {100 * cls['bn'] / tot:.0f} % ({cls['bn']:.1f}; round 3: 14.2 + 3.1 of pool
backward), the pool backward passes that are left {cls['pool']:.1f} ms, BiLSTM {100 * cls['lstm'] / tot:.1f} % ({cls['lstm']:.1f}), C_in = 1 convs + head {cls['c1head']:.1f} ms (incl. the CRNN's
conv1 -> ReLU -> pool backward), abs-max passes {cls['absmax']:.1f}, everything else (CTC, Adam, jitter, decode / edit distance, derived weight forms,
memsets) {cls['other']:.1f}.  The HBM-bound share is {hbm:.1f} ms = {100 * hbm / tot:.0f} % of the step (round 3: 20.3 ms = 19 %).  Box-to-box spread of one build is
± 2.5 %, set by the clock the dominant kernel holds.
<!-- classes:end -->"""
    s = open(P("DESIGN.md")).read()
    a, b = s.index(BEGIN), s.index(END) + len(END)
    s = s[:a] + nums + s[b:]
    if "<!-- classes:begin -->" in s:
        a, b = s.index("<!-- classes:begin -->"), s.index("<!-- classes:end -->") + len("<!-- classes:end -->")
        s = s[:a] + classes + s[b:]
    open(P("DESIGN.md"), "w").write(s)
    s = open(P("profiles", "README.md")).read()
    s = re.sub(r"CROSSCHECK[^|]*\|", f"`{DOM}` {tr[0]} calls (13 steps × {tr[0] // 13}), average {tr[1]:.1f} µs in the trace against `roofline.avg_launch_us` = "
               f"{u['roofline']['avg_launch_us']:.1f} µs from bench.py's HIP events in the same run |", s, count=1)
    open(P("profiles", "README.md"), "w").write(s)
    print("ok", d["value"], r["frac"], tr, u["roofline"]["avg_launch_us"])


if __name__ == "__main__":
    main()
