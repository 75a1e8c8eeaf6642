"""Which candidates of tests/golden/cond_b*.npz are free of decision flips for THIS build of the HIP path, per MFMA mode.

    python tools/qualify_fixtures.py          # on the GPU box; writes gpurun_out/cond_qualified.json (copy to tests/golden/)

A candidate is 'qualified' for a mode when every gradient tensor of both phases is within 1e-4 (plain full-tensor
l2-relative error) of the fp64 oracle.  Candidates were selected on the CPU because nine fp32 evaluations of the reference
agree with its fp64 run on them; whether an INDEPENDENT fp32 implementation (other summation orders) also takes every
ReLU / max-pool decision the same way can only be established by running it.  The strict test
(tests/test_conditioned_gpu.py::test_qualified_candidates_within_1e4) holds the qualified ones to 1e-4 from then on; the
statistical test over ALL candidates does not depend on this file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import helpers as H  # noqa: E402
import cond_runner as cr  # noqa: E402
from qea import ops  # noqa: E402

CASES = ["cond_b2w32.npz", "cond_b4w64.npz", "cond_b4w128.npz"]


def main():
    out = {}
    for case in CASES:
        fx = H.golden(case)
        out[case] = {}
        for mode in ("split_bf16", "f32"):
            prev = ops.set_mfma_mode(mode)
            try:
                ok = []
                for ci in range(int(fx["n_candidates"])):
                    r = cr.run_candidate(case, fx, f"c{ci}|")
                    clean = r["worst"] <= 1e-4 and r["worst_direct"] <= 1e-4
                    print(f"{case} {mode} c{ci}: worst {r['worst']:.2e} ({r['worst_tag']}) median {r['median']:.2e} loss {r['loss_B']:.1e}/{r['loss_A']:.1e}"
                          f" -> {'qualified' if clean else 'decision flip'}", flush=True)
                    if clean:
                        ok.append(ci)
                out[case][mode] = ok
            finally:
                ops.set_mfma_mode(prev)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cond_qualified.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
