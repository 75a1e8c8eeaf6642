"""Reads hipcc's -Rpass-analysis=kernel-resource-usage remarks (obj/*.res, written by build.sh): prints the kernels that use
scratch memory (register spills) and fails when one exceeds QEA_MAX_SCRATCH bytes per lane."""
import os
import re
import subprocess
import sys

LIMIT = int(os.environ.get("QEA_MAX_SCRATCH", "512"))


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return [re.sub(r"\(anonymous namespace\)::", "", l.split("(")[0]) for l in out.splitlines()]
    except (OSError, subprocess.CalledProcessError):
        return names


rows = []
for path in sys.argv[1:]:
    cur = None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "file": os.path.basename(path)[:-4]}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"remark:\s+VGPRs: (\d+)"), ("agpr", r"remark:\s+AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
spill = [r for r in rows if r.get("scratch", 0) > 0]
names = demangle([r["name"] for r in spill])
bad = []
for r, n in zip(spill, names):
    print(f"[resources] {r['file']}: {n}: {r['scratch']} B/lane scratch, {r.get('vgpr')} VGPRs + {r.get('agpr')} AGPRs, occupancy {r.get('occ')}")
    if r["scratch"] > LIMIT:
        bad.append(n)
print(f"[resources] {len(rows)} kernels, {len(spill)} with scratch, limit {LIMIT} B/lane")
if bad:
    sys.exit(f"resource_report: scratch above {LIMIT} B/lane in {bad}")
