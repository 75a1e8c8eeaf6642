import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "micro"))
import lstm_seq_bench as lb
for B in (1024, 2048):
    a = lb.run(31, B, "seq", reps=10)
    print(os.environ.get("QEA_HIP_LIB", "default")[-12:], B, "fwd %.1f bwd %.1f" % (a["fwd"], a["bwd"]), "nan", bool(torch.isnan(a["bwd_res"]).any()), flush=True)
