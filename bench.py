"""Headline benchmark: patch-images/sec through the preprocessor-training inner loop (BASELINE.json metric) on N MI355X
GPUs of one node.  Default workload = BASELINE configs[2]: per-GPU minibatch of 2048 synthetic 32x128 grey patches, the
FULL minibatch step of train_nn_area.py:212-287:

  Phase A: UNet(eval, no grad) -> TopKCER pick (minibatch_subset_prop 0.95: k = 5 % of the minibatch; whole-minibatch ranking
           over the ranks when N > 1) -> inner_limit = 4 jitter replicas (one Philox launch, replicas fused in the batch dim)
           -> CRNN(train-mode BN per replica group) -> CTC -> backward of the last replica -> [RCCL all-reduce of the flat
           CRNN gradient] -> fused Adam(CRNN)
  Phase B: UNet(train BN) -> CRNN(train, BN eval) -> CTC(mean) + MSE(img, 1) -> backward (UNet dgrad+wgrad, CRNN dgrad+wgrad:
           reference-faithful, 9.846 GFLOP/img) -> [RCCL all-reduce of the flat UNet gradient] -> fused Adam(UNet)
  CER:     greedy decode of Phase B's log-probs -> per-sample edit distance / len -> sampler.update_cer (train_nn_area.py:290-304;
           decode and distances on the device, one [B] int32 copy back) — the table the NEXT step's TopKCER ranks

`value` = minibatch images / time of (Phase A + Phase B + CER update), whole job; `full_step_without_cer_update` keeps the
round-2 definition (the step cut at train_nn_area.py:287) beside it.  Secondary objects on the same line: `phase_b`
(Phase-B-only rate at the same batch: the per-image unit of SURVEY.md §8d) and `configs1_b512` (BASELINE configs[1]: B = 512,
Phase B).  `--phase-b-only` makes the Phase-B rate the `value` instead.

Inputs are resident in HBM, random-init weights, fp32 storage.  One JSON line on rank 0 (contract in the task statement), plus
  "roofline":     the implicit-GEMM MFMA conv class (dominant): algorithmic flops / HIP-event time over the same K steps
                  re-run single-stream; and the SAME class with every product forced onto the fp32 MFMA instruction
                  (`native_fp32`), against the 157.3 TFLOP/s fp32 matrix peak
  "cpu_baseline": the CPU oracle (oracle/, a torch-CPU restatement pinned to the reference) timed on the host cores on a
                  bounded sample at B = 32 and B = 128 — N = 1 only.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu] [--phase-b-only]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL between ranks; must be set before HIP starts

import torch  # noqa: E402

CHARS = 95
FLOP_PER_IMG_FAITHFUL = 9.846e9      # 3 x 3.2819 GFLOP (dgrad + wgrad everywhere)
FP32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md, fp32-input MFMA (= vector peak)
BF16_MFMA_PEAK_TFLOPS = 2516.6       # MI355X_MICROARCH.md, dense bf16 MFMA (256 CUs x 2048 MAC/clk x 2.4 GHz)
SPLIT_BF16_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6   # fp32-equivalent: six bf16 MFMAs per fp32 multiply-add (h+m+l split)
SPLIT_F16_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3    # fp32-equivalent: three fp16 MFMAs per fp32 multiply-add (scaled h+l split; fp16 dense peak = bf16's)


def synth_batch(B, seed, device):
    g = torch.Generator().manual_seed(seed)
    m = (torch.rand(B, 1, 32, 128, generator=g) < 0.12).float()
    ink = torch.rand(B, 1, 32, 128, generator=g) * 0.7 + 0.3
    x = (1 - m * ink + 0.02 * torch.randn(B, 1, 32, 128, generator=g)).clamp(0, 1)
    lens = torch.randint(1, 13, (B,), generator=g)
    y = torch.randint(1, CHARS, (int(lens.sum()),), generator=g).to(torch.int32)
    return x.to(device), y, lens.to(torch.int32)


def host_cpu():
    """What the CPU baseline runs on: model name, logical CPUs, physical cores (unique (package, core) pairs of
    /proc/cpuinfo), CPUs this process may run on."""
    info = {"logical_cpus": os.cpu_count()}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["affinity_cpus"] = os.cpu_count()
    try:
        cores, model, phys, core = set(), None, None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        info["model"] = model
        info["physical_cores"] = len(cores) or None
    except OSError:
        pass
    return info


def cpu_threads(host):
    """threads for the CPU leg: the physical cores this process may use; a 1-GPU job on the GPU box has a 16-CPU share
    whatever os.cpu_count() says, and OpenMP oversubscribed far beyond the share crawls (QEA_CPU_THREADS overrides)."""
    if os.environ.get("QEA_CPU_THREADS"):
        return max(1, int(os.environ["QEA_CPU_THREADS"]))
    n = host.get("affinity_cpus") or 1
    if host.get("physical_cores"):
        n = min(n, host["physical_cores"])
    return max(1, min(n, 16))


def cpu_baseline_one(batch, steps, cores):
    """Phase-B step of the CPU oracle on the host cores (bounded sample)."""
    import torch.nn.functional as F
    from oracle import model_oracle as mo
    torch.set_num_threads(cores)
    x, y, lens = synth_batch(batch, 7, "cpu")
    Pu, Bu = mo.split_state(mo.seeded_state(mo.unet_state_shapes(), 1))
    Pc, Bc = mo.split_state(mo.seeded_state(mo.crnn_state_shapes(), 2))
    opt = torch.optim.Adam(list(Pu.values()), lr=5e-5)
    ins = torch.full((batch,), 31, dtype=torch.int32)

    def step():
        for p in list(Pu.values()) + list(Pc.values()):
            p.grad = None
        img = mo.unet_forward(Pu, Bu, x, training=True)
        lp = mo.crnn_forward(Pc, Bc, img, bn_training=False)
        loss = F.ctc_loss(lp, y, ins, lens) + F.mse_loss(img, torch.ones_like(img))
        loss.backward()
        opt.step()

    print(f"[bench] cpu_baseline: {cores} threads, B={batch}", file=sys.stderr, flush=True)
    step()
    times = []
    for i in range(steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline step {i + 1}/{steps} {times[-1]:.2f}s", file=sys.stderr, flush=True)
    med = sorted(times)[len(times) // 2]
    # the host is shared with the pool's other jobs: the MEDIAN step is the figure (a neighbour's burst lands in one or two steps)
    return {"value": batch / med, "unit": "patch-images/s", "ms_per_step": med * 1e3, "batch": batch, "steps": steps,
            "ms_per_step_min": min(times) * 1e3, "ms_per_step_max": max(times) * 1e3, "ms_per_step_mean": sum(times) / steps * 1e3}


def cpu_baseline():
    """B = 32 and B = 128, the node's core count stated.  ~25 s of CPU work in all.  `value` is the better of the two
    batch sizes (per image): on the pool's hosts the B = 128 working set (67 MB per level-1 tensor) leaves the L3 slice of the 16
    cores a 1-GPU slot owns and runs from DRAM shared with seven other slots, so B = 32 is the faster CPU configuration there."""
    host = host_cpu()
    cores = cpu_threads(host)
    b128 = cpu_baseline_one(128, 6, cores)
    b32 = cpu_baseline_one(32, 8, cores)
    best = b128 if b128["value"] >= b32["value"] else b32
    return {"value": best["value"], "unit": "patch-images/s", "cores": cores, "kind": "port", "batch": best["batch"],
            "cores_note": f"{cores} threads = the CPU share of one GPU slot of this pool (the box refuses larger worker pools); the host has "
                          f"{host.get('physical_cores')} physical cores / {host.get('logical_cpus')} logical CPUs shared by all slots",
            "sample": f"Phase-B steps (UNet train-BN -> CRNN -> CTC+MSE -> backward -> Adam) of the CPU oracle, torch {torch.__version__} CPU fp32: "
                      f"6 steps of B=128 and 8 steps of B=32, 1 warm-up each, median step time; value = the faster batch size per image",
            "b128": b128, "b32": b32, "host": host}


def small_batch_legs():
    """The reference's own batch regime (area_cli.py:11 --batch_size 32; one document of ~20 strips in the patch flow): the Phase-B step
    at B = 32 and B = 8, eager and as ONE hipGraph replay (area_cli's [new] --graph), each in a child process of this bench
    (the numbers a maintainer who runs the README command gets)."""
    import subprocess
    res = {"note": "Phase-B step (UNet train-BN -> CRNN -> CTC + MSE -> backward -> Adam) at the reference's batch sizes; ms per step; "
                   "'graph' = one hipGraph replay per step (area_cli --graph), 'eager' = ~600 launches from Python"}
    for b in (32, 8):
        for mode, flag in (("graph", ["--graph"]), ("eager", [])):
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", str(b), "--phase-b-only", "--steps", "40", "--warmup", "3",
                                    "--no-cpu-baseline", "--no-secondary"] + flag, capture_output=True, text=True, timeout=180)
                d = json.loads(p.stdout.strip().splitlines()[-1])
                res[f"b{b}_{mode}_ms"] = round(d["ms_per_step"], 3)
            except Exception as e:                          # a failed leg must not cost the headline line
                res[f"b{b}_{mode}_ms"] = None
                res.setdefault("errors", []).append(f"B={b} {mode}: {type(e).__name__}")
    return res


DOMINANT_KERNEL = "conv3x3_halo_m16_kernel<128, false, 0, 0, false>"   # (round 4: the 16x16x32 form of the two-way fp16 LDS-halo conv, 64-channel chunks)
DOMINANT_KERNEL_BF16 = "conv3x3_halo_bf3_kernel<64, 128, 4, false, 0, 3>"


def source_hash():
    """sha256 (16 hex digits) over the kernel sources: stamps PMC profiles so that a stale one is refused."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "qea_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048, help="patches per GPU (weak scaling); BASELINE configs[2] = 2048, configs[1] = 512")
    ap.add_argument("--phase-b-only", action="store_true", help="`value` = Phase-B-only rate (no Phase A in the timed region)")
    ap.add_argument("--full-step", action="store_true", help="(default; kept for compatibility) `value` = Phase A + Phase B")
    ap.add_argument("--inner-limit", type=int, default=4)
    ap.add_argument("--skip-crnn-wgrad", action="store_true",
                    help="skip the CRNN weight gradients the reference computes but discards when --update_CRNN is off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] (B = 512) and native-fp32 legs")
    ap.add_argument("--no-phase-b-leg", action="store_true",
                    help="skip the Phase-B-only leg too: every step of the run is then the same full step (rocprofv3 runs: the trace's "
                         "per-kernel averages are then directly comparable with roofline.avg_launch_us)")
    ap.add_argument("--graph", action="store_true",
                    help="record the Phase-B step into a hipGraph and time replays (single GPU, implies --phase-b-only)")
    args = ap.parse_args()
    if args.graph:
        args.phase_b_only = True

    if args.gpus > 1 and "RANK" not in os.environ:
        # convenience: `python bench.py --gpus N` starts the N ranks itself (as a child, before this process touches the GPU)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    # stdout carries exactly ONE line, the JSON: libraries that print banners to fd 1 (RCCL prints its version block there when a
    # communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU with torch.distributed.run")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path for the product)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # under torch.distributed.run (RANK set) the RCCL group is created even for one rank, so the
    # collective path is exercised on a 1-GPU box exactly as it runs on 8
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from qea import dist as qdist
    from qea import ops
    from qea.loss import CTCLoss
    from qea.optim import FusedAdam
    from qea.params import ensure_flat
    import properties
    from selection_utils import datasampler_factory
    from transform_helper import AddGaussianNoice
    from utils import batch_cers, get_char_maps
    c2i, i2c, _ = get_char_maps(properties.char_set)

    torch.manual_seed(42)
    prep = UNet().to(dev)
    crnn = CRNN(CHARS, False).to(dev)
    crnn.register_backward_hook(crnn.backward_hook)
    if args.skip_crnn_wgrad:
        crnn.__dict__["_qea_skip_param_grads"] = True
    if args.graph and use_dist:
        raise SystemExit("bench.py: --graph is a single-GPU option")
    opt_p = FusedAdam(prep.parameters(), lr=5e-5, weight_decay=0, capturable=args.graph)
    opt_c = FusedAdam(crnn.parameters(), lr=1e-4, weight_decay=0)
    ctc = CTCLoss()
    mse = torch.nn.MSELoss()
    fs, fc = ensure_flat(prep), ensure_flat(crnn)
    R = args.inner_limit

    class Work:
        """one per-GPU minibatch of B patches with its labels, CER table and Phase-A label set"""

        def __init__(self, B):
            self.B = B
            self.x, self.y, self.lens = synth_batch(B, 1000 + rank, dev)
            self.ins = torch.full((B,), 31, dtype=torch.int32)
            self.ones = torch.ones(B, 1, 32, 128, device=dev)
            self.names = [f"r{rank}s{i}" for i in range(B)]
            gen = torch.Generator().manual_seed(7 + rank)
            self.sampler = datasampler_factory("topKCER")({n: float(c) for n, c in zip(self.names, torch.rand(B, generator=gen))})
            self.k_global = max(1, -(-B * world * 5 // 100))         # ceil(0.05 * global minibatch): minibatch_subset_prop 0.95
            self.off = torch.zeros(B + 1, dtype=torch.int64)
            self.off[1:] = torch.cumsum(self.lens.to(torch.int64), 0)
            yl = self.y.tolist()
            self.labels = ["".join(i2c[t] for t in yl[int(self.off[i]):int(self.off[i + 1])]) for i in range(B)]
            if args.graph:                                           # capturable form: device-resident targets
                ctc.max_target_length = int(self.lens.max())
                self.y_s, self.ins_s, self.lens_s = self.y.to(dev), self.ins.to(dev), self.lens.to(dev)
            else:
                self.y_s, self.ins_s, self.lens_s = self.y, self.ins, self.lens

    noiser = AddGaussianNoice(std=5, is_stochastic=True)

    # self-diagnosis of the data-parallel legs: HIP events on the compute stream around the two gradient all-reduces
    # (the stream waits for the collective, so the pair brackets it), the rows each rank ran through Phase A after the re-balance
    dp_stats = {"events": {"unet": [], "crnn": []}, "phase_a_rows": 0}

    def timed_allreduce(buf, which):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.all_reduce(buf)
        e1.record()
        dp_stats["events"][which].append((e0, e1))

    def cer_update(w):
        """train_nn_area.py:290-304: pred_to_string -> compare_labels per sample -> sampler.update_cer"""
        w.sampler.update_cer(batch_cers(w.last_lp, w.labels, c2i), w.names)

    def phase_b(w):
        prep.train()
        crnn.train()
        for m in crnn.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
        prep.zero_grad()
        crnn.zero_grad()
        img = prep(w.x)
        lp = crnn(img)
        loss = ctc(lp, w.y_s, w.ins_s, w.lens_s) + mse(img, w.ones)
        w.last_lp = lp.detach()
        loss.backward()
        if use_dist:
            timed_allreduce(fs.grad, "unet")         # one RCCL all-reduce of the flat 31 MB UNet gradient
            if world > 1:
                fs.grad.mul_(1.0 / world)
        opt_p.step()
        return loss

    def phase_a(w, select_first=False):
        """train_nn_area.py:214-275.  The black-box OCR itself is outside the path: its labels are the (fixed) ground truth.
        select_first (--select_before_clean of area_cli, NOT the headline): TopKCER ranks names / CERs only, so the pick can
        precede the cleaner pass, which then runs on the k picked images instead of the whole minibatch."""
        crnn.train()
        prep.eval()
        prep.zero_grad()
        crnn.zero_grad()
        if select_first:
            preds_all = w.x
        else:
            with torch.no_grad():
                preds_all = prep(w.x)
        if world > 1:                                # whole-minibatch ranking over the ranks (32 KB all-gather of the CERs)
            preds, _, idx, kg, counts = w.sampler.query_global(preds_all, w.names, w.k_global, w.names, with_counts=True)
        else:
            preds, _, idx = w.sampler.query(preds_all, w.names, w.k_global, w.names)
            share = 1.0
        if select_first and preds.shape[0]:
            with torch.no_grad():
                preds = prep(preds.contiguous())
        if world > 1:
            # the global winners dealt out again in equal slices (train_nn_area's default under DP): one all-reduce of k x 16 KB;
            # the synthetic label of a strip that came from another rank is one of this rank's own (labels are fixed stand-ins
            # for the black box's answers either way)
            preds = qdist.rebalance_rows(preds.contiguous(), counts)
            idx = torch.arange(preds.shape[0])
            share = world * preds.shape[0] / kg
        k = preds.shape[0]
        if k:
            # all replicas in ONE Philox launch and ONE CRNN pass with per-replica-group BatchNorm
            noisy, _ = noiser.batch(preds, replicas=R)
            lpA = crnn(noisy, replica_groups=R, backward_group=R - 1)   # as train_nn_area does: the last replica's samples only
            sel = idx.tolist()
            yA = torch.cat([w.y[int(w.off[i]):int(w.off[i + 1])] for i in sel])
            lossA = ctc(lpA[:, (R - 1) * k:, :], yA, torch.full((k,), 31, dtype=torch.int32), w.lens[idx.cpu()])
            (lossA * share if share != 1.0 else lossA).backward()      # area flow: last replica only
        dp_stats["phase_a_rows"] = k                 # rows this rank ran through Phase A (after the winner re-balance)
        if use_dist:
            timed_allreduce(fc.grad, "crnn")         # flat 35 MB CRNN gradient; the second all-reduce of the step
            if world > 1:
                fc.grad.mul_(1.0 / world)
        opt_c.step()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item(), out

    W = Work(args.batch)
    B = W.B

    def full_step():
        phase_a(W)
        loss = phase_b(W)
        cer_update(W)
        return loss

    def full_step_r2():                              # the round-2 definition: cut at train_nn_area.py:287
        phase_a(W)
        return phase_b(W)

    main_step = (lambda: phase_b(W)) if args.phase_b_only else full_step
    for i in range(args.warmup):
        main_step()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[bench] warm-up {i + 1}/{args.warmup} done", file=sys.stderr, flush=True)
    # ---- timed region: the production configuration (weight gradients overlapped on a side stream), no event overhead
    run = main_step
    if args.graph:
        from qea.graph import GraphedStep
        run = GraphedStep(lambda: phase_b(W), warmup=0)
        run()
    for v in dp_stats["events"].values():
        v.clear()                                    # only the timed region's collectives are reported
    dt, loss = timed(run, args.steps)
    dp_report = None
    if use_dist:
        torch.cuda.synchronize()
        ar = {k: sum(a.elapsed_time(b) for a, b in v) / max(args.steps, 1) for k, v in dp_stats["events"].items()}
        rows = [None] * world
        dist.all_gather_object(rows, int(dp_stats["phase_a_rows"]))
        dp_report = {"rccl_ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                     "allreduce_ms": {"crnn_after_phase_a": ar["crnn"], "unet_after_phase_b": ar["unet"], "per_step": ar["crnn"] + ar["unet"],
                                      "note": "HIP events on the compute stream around dist.all_reduce, mean per timed step on rank 0"},
                     "phase_a_rows_per_rank": rows}
        for v in dp_stats["events"].values():
            v.clear()
    if rank == 0:
        print(f"[bench] {args.steps} timed steps in {dt:.3f}s", file=sys.stderr, flush=True)
    # ---- Phase B alone at the same batch (the per-image unit of SURVEY.md §8d)
    dt_b = dt
    if not args.phase_b_only and not args.no_phase_b_leg:
        phase_b(W)
        dt_b, _ = timed(lambda: phase_b(W), args.steps)

    # ---- roofline leg: the same K steps once more with the side stream off and every MFMA launch bracketed by HIP
    # events on its stream — with the overlap on, a launch's event-to-event time would include a co-running kernel
    def event_leg(fn, steps):
        overlap0 = ops.overlap_enabled()             # QEA_OVERLAP=0 keeps the whole run single-stream (profiles/)
        ops.set_overlap(False)
        fn()
        for k in (ops.PROF_CONV_IGEMM, ops.PROF_CONV_WGRAD, ops.PROF_LSTM_STEP):
            ops.prof_enable(k, True)
        ops.prof_reset()
        d, _ = timed(fn, steps)
        prof = {k: ops.prof_read(k) for k in (ops.PROF_CONV_IGEMM, ops.PROF_CONV_WGRAD, ops.PROF_LSTM_STEP)}
        for k in list(prof):
            ops.prof_enable(k, False)
        # the single dominant kernel of the step (rocprofv3 lists it as conv3x3_halo_m16_kernel<128, false, 0, 0, false>; bf16 mode: conv3x3_halo_bf3_kernel<64, 128, 4, false, 0, 3>)
        prof["dominant"] = ops.prof_read_tagged(ops.PROF_CONV_IGEMM, ops.prof_tag_halo_bf3(64, 128, False, f16=ops.mfma_mode() == "split_f16"))
        ops.set_overlap(overlap0)
        return prof, d, overlap0

    prof, dt_serial, overlap0 = event_leg(main_step, args.steps)
    native = None
    if not args.no_secondary:
        # the same steps with every product on v_mfma_f32_32x32x2_f32: the class against the fp32 matrix peak
        prev = ops.set_mfma_mode("f32")
        try:
            nsteps = max(2, min(args.steps, 4))
            nprof, ndt, _ = event_leg(main_step, nsteps)
            native = (nprof, ndt, nsteps)
        finally:
            ops.set_mfma_mode(prev)
    bf16_leg = None
    if not args.no_secondary and ops.mfma_mode() == "split_f16":
        # the same steps in the round-2 form (three-way bf16 split, six MFMAs per product): the two forms on the same box
        prev = ops.set_mfma_mode("split_bf16")
        try:
            nsteps = max(2, min(args.steps, 4))
            main_step()
            bprof, bdt, _ = event_leg(main_step, nsteps)
            bf16_leg = (bprof, bdt, nsteps)
        finally:
            ops.set_mfma_mode(prev)
    no_cer = None
    if not args.no_secondary and not args.phase_b_only:
        full_step_r2()
        dnc, _ = timed(full_step_r2, args.steps)
        no_cer = {"note": "Phase A + Phase B only (the round-2 timed region: the step cut at train_nn_area.py:287, before decode -> CER -> update_cer)",
                  "value": B * world * args.steps / dnc, "unit": "patch-images/s", "ms_per_step": dnc / args.steps * 1e3}
    sel_first = None
    if not args.no_secondary and not args.phase_b_only:
        def step_sf():
            phase_a(W, select_first=True)
            loss = phase_b(W)
            cer_update(W)
            return loss
        step_sf()
        dsf, _ = timed(step_sf, args.steps)
        sel_first = {"note": "the same full step with area_cli's [new] --select_before_clean: TopKCER picks on names / CERs, the eval-mode "
                             "cleaner then runs on the k picked images only (same results up to rounding; NOT the reference's order of work)",
                     "value": B * world * args.steps / dsf, "unit": "patch-images/s", "ms_per_step": dsf / args.steps * 1e3}
    c1 = None
    if not args.no_secondary and B != 512:
        W1 = Work(512)
        for _ in range(2):
            phase_b(W1)
        d1, _ = timed(lambda: phase_b(W1), args.steps)
        c1 = {"workload": "BASELINE configs[1]: Phase-B step, B = 512 per GPU", "value": 512 * world * args.steps / d1, "unit": "patch-images/s",
              "ms_per_step": d1 / args.steps * 1e3}
        del W1

    traffic, traffic_dom, traffic_note = None, None, "no PMC profile for this build"
    try:                                             # HBM bytes per launch of the dominant class, from the committed PMC pass of THIS build
        pm = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))
        if pm.get("source_hash") != source_hash():
            traffic_note = f"profiles/r04_pmc_traffic.json was taken on sources {pm.get('source_hash')} != this build {source_hash()}: refused"
        elif pm.get("batch_per_gpu") != B or pm.get("full_step") != (not args.phase_b_only):
            traffic_note = "profiles/r04_pmc_traffic.json was taken on another workload: refused"
        else:
            traffic = pm["conv_igemm"]["hbm_bytes_per_launch"]
            kt = pm.get("per_kernel_bytes_per_launch", {}).get(DOMINANT_KERNEL)
            traffic_dom = kt["fetch"] + kt["write"] if kt else None
            traffic_note = "HBM bytes/launch, rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE in separate passes, profiles/r04_pmc_traffic.json (same sources)"
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        ig, wg, ls = prof[ops.PROF_CONV_IGEMM], prof[ops.PROF_CONV_WGRAD], prof[ops.PROF_LSTM_STEP]
        tf = lambda q: q["flops"] / (q["ms"] * 1e-3) / 1e12 if q["ms"] > 0 else 0.0
        ach = tf(ig)
        dom = prof["dominant"]
        dom_tf = tf(dom)
        # the class mixes split-bf16 launches (>= 128-channel layers) and native fp32-MFMA launches: its matrix roofline is
        # the flop-weighted harmonic blend of the two peaks (time at peak = flops_split / peak_split + flops_f32 / peak_f32)
        # time at peak = flops_f16 / peak_f16split + flops_bf16 / peak_bf16split + flops_fp32 / peak_fp32
        blend = lambda q: 1.0 / ((q["flops_split_f16"] / q["flops"]) / SPLIT_F16_PEAK_TFLOPS + (q["flops_split_bf16"] / q["flops"]) / SPLIT_BF16_PEAK_TFLOPS
                                 + (1.0 - (q["flops_split_bf16"] + q["flops_split_f16"]) / q["flops"]) / FP32_MFMA_PEAK_TFLOPS) if q["flops"] > 0 else FP32_MFMA_PEAK_TFLOPS
        f_split = ig["flops_split_bf16"] / ig["flops"] if ig["flops"] > 0 else 0.0
        f_f16 = ig["flops_split_f16"] / ig["flops"] if ig["flops"] > 0 else 0.0
        f16_mode = ops.mfma_mode() == "split_f16"
        dom_peak = SPLIT_F16_PEAK_TFLOPS if f16_mode else SPLIT_BF16_PEAK_TFLOPS
        peak = blend(ig)
        imgs = B * world * args.steps
        workload = ("BASELINE configs[2]: full minibatch step of train_nn_area.py:212-287 — Phase A (UNet eval fwd, TopKCER k = 5 % of the "
                    f"minibatch, inner_limit = {R} jitter replicas fused in the batch dim, CRNN train-BN fwd+bwd, Adam(CRNN)) + Phase B (UNet "
                    "train-BN -> CRNN BN-eval -> CTC mean + MSE -> backward -> Adam(UNet)) + greedy decode -> CER -> update_cer (:290-304) "
                    "on synthetic POS-style 32x128 patches"
                    if not args.phase_b_only else
                    "Phase-B step (UNet train-BN -> CRNN BN-eval -> CTC mean + MSE -> backward -> Adam(UNet)) on synthetic POS-style 32x128 patches")
        out = {
            "metric": "patch-images/sec UNet->CRNN->CTC fwd+bwd, 32x128 grey",
            "value": imgs / dt,
            "unit": "patch-images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "split_bf16": "f32 (GEMMs as 3 x bf16 split products, fp32 accumulate)",
                      "split_f16": "f32 (GEMMs as scaled 2 x fp16 split products — three fp16 MFMAs per fp32 multiply-add — fp32 accumulate)"}[ops.mfma_mode()],
            "data": "synthetic",
            "overlap": {"wgrad_side_stream": overlap0, "ms_per_step_single_stream": dt_serial / args.steps * 1e3},
            "config": {"workload": workload, "batch_per_gpu": B, "global_batch": B * world, "full_step": not args.phase_b_only,
                       "phase_a": None if args.phase_b_only else {
                           "selection": "topKCER", "minibatch_subset_prop": 0.95, "k_global": W.k_global, "inner_limit": R,
                           "ranking": "whole minibatch over the ranks (all-gather of the CERs)" if world > 1 else "whole minibatch",
                           "backward": "last replica (train_nn_area.py:269-271)", "replicas": "fused in the batch dim, per-group BN",
                           "ocr": "fixed labels (black box excluded)"},
                       "crnn_wgrad": not args.skip_crnn_wgrad, "parallelism": f"dp{world}",
                       "collectives_per_step": 0 if not use_dist else (1 if args.phase_b_only else 2),
                       "small_collectives_per_step": 0 if (world == 1 or args.phase_b_only) else "CER all-gather (4 B per strip) + winner re-balance (k x 16 KB)",
                       "data_parallel": dp_report,
                       "loss": float(loss.item()), "hipgraph": bool(args.graph)},
            "phase_b": None if (args.no_phase_b_leg and not args.phase_b_only) else {"value": imgs / dt_b, "unit": "patch-images/s", "ms_per_step": dt_b / args.steps * 1e3,
                        "end_to_end_tflops": FLOP_PER_IMG_FAITHFUL * imgs / dt_b / 1e12 if not args.skip_crnn_wgrad else None,
                        "note": "Phase B alone at the same batch: 9.846 GFLOP per image (SURVEY.md §8d unit of work)"},
            "roofline": {"bound": "mfma",
                         "kernel": (DOMINANT_KERNEL if f16_mode else DOMINANT_KERNEL_BF16) + " — the LDS-halo 3x3 convolution (conv_igemm.hip): fp32 "
                                   "operands scaled by the power of two of their abs-max and split into 2 fp16 planes (input halo split once per tile "
                                   "in LDS, filter fragments pre-split), three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate [split_bf16 mode: "
                                   "3 bf16 planes, six MFMAs]; forward and input gradient of every 3x3 layer with >= 64 input and >= 128 output "
                                   "channels and W % 32 == 0",
                         "achieved": dom_tf, "peak": dom_peak, "unit": "TFLOP/s", "frac": dom_tf / dom_peak,
                         "frac_of_the_six_mfma_peak": dom_tf / SPLIT_BF16_PEAK_TFLOPS,
                         "avg_launch_us": dom["ms"] * 1e3 / max(1, dom["launches"]), "launches_per_step": dom["launches"] / args.steps,
                         "ms_per_step_in_kernel": dom["ms"] / args.steps, "share_of_step": dom["ms"] / args.steps / (dt_serial / args.steps * 1e3),
                         "algorithmic_flops_per_launch": dom["flops"] / max(1, dom["launches"]),
                         "algorithmic_bytes_per_launch": dom["bytes"] / max(1, dom["launches"]),
                         "traffic": traffic_dom, "traffic_note": traffic_note, "source_hash": source_hash(),
                         "measured": "HIP events around every launch of this kernel over the same K steps re-run with the wgrad side stream "
                                     f"disabled ({dt_serial / args.steps * 1e3:.2f} ms/step single-stream vs {dt / args.steps * 1e3:.2f} overlapped); "
                                     "avg_launch_us is comparable with profiles/r03_kernel_stats_b2048_single_stream.csv",
                         "peak_is": (f"fp16 dense MFMA peak / 3 = {SPLIT_F16_PEAK_TFLOPS:.1f} TFLOP/s fp32-equivalent (three MFMAs per product); rounds 1-2 priced "
                                     f"the six-MFMA bf16 form against {SPLIT_BF16_PEAK_TFLOPS:.1f} (frac_of_the_six_mfma_peak)") if f16_mode else
                                    f"bf16 dense MFMA peak / 6 = {SPLIT_BF16_PEAK_TFLOPS:.1f} TFLOP/s fp32-equivalent",
                         "launch_class": {
                             "what": "all qea_conv_igemm launches (implicit-GEMM conv fwd/dgrad, convT, LSTM/linear GEMMs): the LDS-halo kernel above, "
                                     "conv_igemm_bf3w_kernel (other >= 128-channel GEMMs: pre-split filter planes by LDS-DMA), and the native "
                                     "v_mfma_f32_32x32x2_f32 kernels conv_igemm_kernel / conv3x3_halo_kernel",
                             "achieved": ach, "peak": peak, "frac": ach / peak, "traffic": traffic,
                             "algorithmic_bytes_per_launch": ig["bytes"] / max(1, ig["launches"]), "launches_per_step": ig["launches"] / args.steps,
                             "ms_per_step_in_kernel": ig["ms"] / args.steps, "split_bf16_flop_fraction": f_split, "split_f16_flop_fraction": f_f16,
                             "frac_of_native_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                             "peak_note": f"fp32-equivalent; flop-weighted blend of fp16 dense peak / 3 = {SPLIT_F16_PEAK_TFLOPS:.1f} ({100 * f_f16:.0f} % of the "
                                          f"class's flops), bf16 dense peak / 6 = {SPLIT_BF16_PEAK_TFLOPS:.1f} ({100 * f_split:.0f} %) and the fp32 MFMA peak "
                                          f"{FP32_MFMA_PEAK_TFLOPS}"}},
            "kernels": {
                "conv_wgrad": {"tflops": tf(wg), "ms_per_step": wg["ms"] / args.steps, "peak": blend(wg), "frac": tf(wg) / blend(wg),
                               "split_bf16_flop_fraction": wg["flops_split_bf16"] / wg["flops"] if wg["flops"] > 0 else 0.0,
                               "split_f16_flop_fraction": wg["flops_split_f16"] / wg["flops"] if wg["flops"] > 0 else 0.0,
                               "algorithmic_bytes_per_launch": wg["bytes"] / max(1, wg["launches"]),
                               "launches_per_step": wg["launches"] / args.steps},
                "lstm_step": {"tflops": tf(ls), "ms_per_step": ls["ms"] / args.steps, "launches_per_step": ls["launches"] / args.steps},
            },
        }
        if native is not None:
            nprof, ndt, nsteps = native
            nig, nwg = nprof[ops.PROF_CONV_IGEMM], nprof[ops.PROF_CONV_WGRAD]
            out["roofline"]["native_fp32"] = {
                "note": "the same steps with every product on v_mfma_f32_32x32x2_f32 (qea_set_mfma_mode(QEA_MFMA_F32)), single stream, HIP events",
                "conv_igemm_tflops": tf(nig), "peak": FP32_MFMA_PEAK_TFLOPS, "frac": tf(nig) / FP32_MFMA_PEAK_TFLOPS,
                "conv_wgrad_tflops": tf(nwg), "conv_wgrad_frac": tf(nwg) / FP32_MFMA_PEAK_TFLOPS,
                "ms_per_step_single_stream": ndt / nsteps * 1e3, "value": B * world * nsteps / ndt, "steps": nsteps}
        if bf16_leg is not None:
            bprof, bdt, nsteps = bf16_leg
            big, bwg, bdom = bprof[ops.PROF_CONV_IGEMM], bprof[ops.PROF_CONV_WGRAD], bprof["dominant"]
            out["roofline"]["split_bf16"] = {
                "note": "the same steps with the round-2 three-way bf16 split (six MFMAs per product; QEA_SPLIT=bf16), single stream, HIP events",
                "kernel": DOMINANT_KERNEL_BF16, "achieved": tf(bdom), "peak": SPLIT_BF16_PEAK_TFLOPS, "frac": tf(bdom) / SPLIT_BF16_PEAK_TFLOPS,
                "avg_launch_us": bdom["ms"] * 1e3 / max(1, bdom["launches"]), "conv_igemm_tflops": tf(big), "conv_wgrad_tflops": tf(bwg),
                "ms_per_step_single_stream": bdt / nsteps * 1e3, "value": B * world * nsteps / bdt, "steps": nsteps}
        if no_cer is not None:
            out["full_step_without_cer_update"] = no_cer
        if sel_first is not None:
            out["full_step_select_before_clean"] = sel_first
        if c1 is not None:
            out["configs1_b512"] = c1
        if world == 1 and not use_dist and not args.no_secondary and not args.phase_b_only and not args.graph and B >= 512:
            out["small_batch"] = small_batch_legs()          # (child processes of this bench: not under torch.distributed.run)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
