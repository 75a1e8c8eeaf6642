"""Per-layer forward error, decision flips and decision-conditioned gradient errors of the HIP path on the cond_b*.npz
candidates, next to the reference's own fp32 figures (tests/golden/ladder.json).  GPU box:
    python tools/forward_ladder.py [case ...] > gpurun_out/ladder.txt        (json beside it: gpurun_out/ladder_hip.json)
TEST/ANALYSIS INFRASTRUCTURE: imports the oracle."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd")):
    sys.path.insert(0, p)

import cond_runner as cr  # noqa: E402
import helpers as H  # noqa: E402
from qea import ops  # noqa: E402

cases = sys.argv[1:] or ["cond_b2w32.npz", "cond_b4w64.npz", "cond_b4w128.npz"]
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ladder.json")))
out = {}
for case in cases:
    fx = H.golden(case)
    for mode in ("split_f16", "split_bf16", "f32"):
        prev = ops.set_mfma_mode(mode)
        for ci in range(int(fx["n_candidates"])):
            r = cr.run_candidate(case, fx, f"c{ci}|")
            cpu = ref[case][f"c{ci}"]
            out[f"{case}|{mode}|c{ci}"] = {k: r[k] for k in ("tensor", "free", "flips", "ladder", "worst", "worst_free", "n_flips",
                                                             "worst_flip_units", "loss_A", "loss_B", "img", "lp", "buf")}
            fl = {k: v for k, v in r["flips"].items() if v[0]}
            print(f"== {case} c{ci} {mode}: conditioned worst {r['worst']:.2e} ({r['worst_tag']}) median {r['median']:.2e}; free worst {r['worst_free']:.2e};"
                  f" flips {r['n_flips']} (cpu default {sum(v[0] for v in cpu['default']['flips_B'].values()) + sum(v[0] for v in cpu['default']['flips_A'].values())},"
                  f" nomkldnn {sum(v[0] for v in cpu['nomkldnn']['flips_B'].values()) + sum(v[0] for v in cpu['nomkldnn']['flips_A'].values())})"
                  f" worst units {r['worst_flip_units']:.1f}; img {r['img']:.1e} lp {r['lp']:.1e}", flush=True)
            for k, v in fl.items():
                print(f"     flip {k}: {v[0]} of {v[1]}, worst margin {v[2]:.1f} units")
            if ci == 0:
                for k, e in r["ladder"].items():
                    ph, site = k.split("|", 1)
                    key = {"relu": None}.get(None)
                    pre = site
                    if "relu" in site:
                        head, leaf = site.rsplit(".", 1)
                        pre = ({"5": "convo.batchnorm1", "6": "convo.batchnorm2"}.get(leaf[-1], "convo.conv" + leaf[-1]) if head == "convo"
                               else head + "." + leaf.replace("relu", "norm"))
                    d = cpu["default"]["err_" + ph].get(pre)
                    n = cpu["nomkldnn"]["err_" + ph].get(pre)
                    print(f"     {k:36s} hip {e:.2e}   cpu default {d if d is None else format(d, '.2e')}  nomkldnn {n if n is None else format(n, '.2e')}"
                          f"{'   <<< ' + format(e / n, '.1f') + 'x' if n and e > 2 * n else ''}")
        ops.set_mfma_mode(prev)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ladder_hip.json"), "w"))
