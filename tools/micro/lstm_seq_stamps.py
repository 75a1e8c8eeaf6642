"""In-kernel phase stamps of the one-launch BiLSTM kernels (profiles/r04_lstm_seq_lab.txt).  Needs a DEBUG build of csrc/lstm_seq.hip
that records s_memrealtime per phase into a __device__ array and exports qea_lstm_seq_debug_read (the STAMP(...) patch of round 4,
not part of the library): with the release library this script stops at the missing symbol."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "micro"))
from qea import ops, _lib
import lstm_seq_bench as lb
names = ["wait", "gemm", "(x)", "gate math", "xchg store", "publish", "plain stores"]
for B in (32, 2048):
    lb.run(31, B, "seq", reps=2)
    buf = (C.c_longlong * (2 * 64 * 8))()
    L = _lib.lib()
    L.qea_lstm_seq_debug_read.argtypes = [C.c_void_p]
    print("rc", L.qea_lstm_seq_debug_read(buf))
    lab = ["prologue", "wait", "gemm", "reduce+gates", "xchg", "publish", "plain stores"]
    for kind, nm in ((0, "fwd"), (1, "bwd")):
        acc = [0.0] * 7
        n = 0
        for st in range(5, 28):
            v = [buf[(kind * 64 + st) * 8 + ph] for ph in range(7)]
            prev_end = buf[(kind * 64 + st - 1) * 8 + 6]
            acc[0] += (v[0] - prev_end) / 100.0
            for ph in range(1, 7):
                acc[ph] += (v[ph] - v[ph - 1]) / 100.0
            n += 1
        print(f"B={B} {nm}: " + "  ".join(f"{l}={a / n:.2f}" for l, a in zip(lab, acc)) + f"  total={sum(acc) / n:.2f} us", flush=True)
