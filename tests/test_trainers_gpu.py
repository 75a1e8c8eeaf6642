"""The two trainers end to end on the MI355X HIP path (default backend), synthetic data + stub OCR."""
import json
import os
import sys

import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _args(which, tmp, **over):
    from qea.cli_flags import build_parser
    a = build_parser(which, "").parse_args(["--exp_base_path", str(tmp), "--ocr", "stub", "--epoch", "1"])
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_area_trainer_hip(tmp_path):
    from datasets.synthetic import SyntheticTextAreas
    from train_nn_area import TrainNNPrep
    tr = SyntheticTextAreas(32, seed=1, include_name=True, include_index=True)
    cers_path = tmp_path / "cers.json"
    json.dump({n: (i % 5) / 4 for i, n in enumerate(tr.names)}, open(cers_path, "w"))
    args = _args("a", tmp_path / "exp", batch_size=16, minibatch_subset="topKCER", minibatch_subset_prop=0.75, cers_ocr_path=str(cers_path),
                 inner_limit=2, inner_limit_skip=True, window_size=2)
    t = TrainNNPrep(args, train_set=tr, val_set=SyntheticTextAreas(16, seed=2, include_name=True))
    assert t.device.type == "cuda" and type(t.prep_model).__module__ == "models.model_unet"
    p0 = torch.cat([p.detach().flatten().clone() for p in t.prep_model.parameters()])
    c0 = torch.cat([p.detach().flatten().clone() for p in t.crnn_model.parameters()])
    t.train()
    p1 = torch.cat([p.detach().flatten() for p in t.prep_model.parameters()])
    c1 = torch.cat([p.detach().flatten() for p in t.crnn_model.parameters()])
    assert torch.isfinite(p1).all() and torch.isfinite(c1).all()
    # Adam's first steps move every weight by about lr
    assert 0 < (p1 - p0).abs().max().item() < 5 * 5e-5 * 2 + 1e-6 and 0 < (c1 - c0).abs().max().item() < 5 * 1e-4 * 2 + 1e-6
    assert t.ocr.count_calls == 2 * 4 * 2 + 16
    ck = tmp_path / "exp" / "ckpts"
    prep = [f for f in os.listdir(ck) if f.startswith("Prep_model_0_")]
    assert prep and os.path.exists(ck / "CRNN_model_0")
    m = torch.load(ck / prep[0], weights_only=False)
    assert type(m).__name__ == "UNet" and len(m.state_dict()) == len(t.prep_model.state_dict())


def test_patch_trainer_hip(tmp_path):
    from datasets.synthetic import SyntheticPatches
    from train_nn_patch import TrainNNPrep
    tr = SyntheticPatches(3, seed=1)
    names = []
    for i in range(len(tr)):
        _, boxes, name = tr[i]
        names += TrainNNPrep._strip_names([b["label"] for b in boxes], name)
    cers_path = tmp_path / "cers.json"
    json.dump({n: (i % 3) / 2 for i, n in enumerate(names)}, open(cers_path, "w"))
    args = _args("p", tmp_path / "exp", minibatch_subset="topKCER", minibatch_subset_prop=0.5, cers_ocr_path=str(cers_path), inner_limit=2)
    t = TrainNNPrep(args, train_set=tr, val_set=SyntheticPatches(1, seed=2, include_name=False))
    t.train()
    flat = torch.cat([p.detach().flatten() for p in t.prep_model.parameters()])
    assert torch.isfinite(flat).all()
    ck = tmp_path / "exp" / "ckpts"
    for f in ("CRNN_model_0", "optim_prep_latest", "optim_crnn_latest"):
        assert os.path.exists(ck / f)
    st = torch.load(ck / "optim_prep_latest", weights_only=False)
    assert len(st["state"]) == len(list(t.prep_model.parameters())) and "exp_avg" in st["state"][0]
    assert set(json.load(open(tmp_path / "exp" / "cers" / "all_cers.json")).keys()) == set(names)


def test_patch_trainer_hip_docs_per_step(tmp_path):
    """[new] --docs_per_step 2 on the HIP path: two documents per optimiser step through one cleaner pass (per-document BatchNorm
    groups) and one CRNN pass; half the steps, every strip's CER refreshed, finite weights."""
    from datasets.synthetic import SyntheticPatches
    from train_nn_patch import TrainNNPrep
    tr = SyntheticPatches(4, seed=1)
    names = []
    for i in range(len(tr)):
        _, boxes, name = tr[i]
        names += TrainNNPrep._strip_names([b["label"] for b in boxes], name)
    cers_path = tmp_path / "cers.json"
    json.dump({n: (i % 3) / 2 for i, n in enumerate(names)}, open(cers_path, "w"))
    args = _args("p", tmp_path / "exp", minibatch_subset="topKCER", minibatch_subset_prop=0.5, cers_ocr_path=str(cers_path), inner_limit=2,
                 docs_per_step=2)
    t = TrainNNPrep(args, train_set=tr, val_set=SyntheticPatches(1, seed=2, include_name=False))
    assert len(t.loader_train) == 2
    steps = []
    orig = t._step_prep
    t._step_prep = lambda also_crnn=False: (steps.append(1), orig(also_crnn))[1]
    t.train()
    assert len(steps) == 2
    assert torch.isfinite(torch.cat([p.detach().flatten() for p in t.prep_model.parameters()])).all()
    assert set(t.sampler.all_cers.keys()) == set(names) and all(len(v) == 1 for v in t.sampler.all_cers.values())


def test_patch_trainer_docs_per_step_batches_phase_a(tmp_path):
    """[new] --docs_per_step N on the HIP path: the CRNN side of Phase A runs ONCE for the N documents (ragged BatchNorm groups, one per
    (document, replica)) instead of once per document; with the jitter pinned to a function of the image content, the CRNN gradient that
    the first Adam(CRNN) step consumes equals the one of the per-document loop (the same trainer with the batching switched off)."""
    from datasets.synthetic import SyntheticPatches
    from train_nn_patch import TrainNNPrep
    import transform_helper
    res = {}
    real_batch = transform_helper.AddGaussianNoice.batch

    def pinned(self, imgs, replicas=1, **_k):
        out = torch.cat([torch.clamp(imgs - 0.05 * (r + 1) * torch.sin(37.0 * imgs + r), 0, 1) for r in range(replicas)])
        return out, None
    transform_helper.AddGaussianNoice.batch = pinned
    try:
        for batched in (True, False):
            torch.manual_seed(0)
            tr = SyntheticPatches(4, seed=1)
            args = _args("p", tmp_path / f"exp{int(batched)}", inner_limit=2, docs_per_step=2)
            t = TrainNNPrep(args, train_set=tr, val_set=SyntheticPatches(1, seed=2, include_name=False))
            t._batch_phase_a = batched
            calls, grads = [], []
            orig = t._replica_losses_docs
            t._replica_losses_docs = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            orig_step = t._step_crnn

            def spy(orig_step=orig_step, t=t, grads=grads):
                grads.append(torch.cat([p.grad.detach().flatten().clone() for p in t.crnn_model.parameters()]))
                return orig_step()
            t._step_crnn = spy
            t.train()
            assert (len(calls) > 0) == batched and len(grads) == 2
            res[batched] = grads[0]
    finally:
        transform_helper.AddGaussianNoice.batch = real_batch
    d = (res[True].double() - res[False].double()).norm().item() / res[False].double().norm().item()
    assert d <= 1e-4, d


def test_area_trainer_graph_replays_both_phases(tmp_path):
    """[new] --graph with Phase A on: the CRNN side of Phase A (forward on the R jittered copies, CTC of the last copy, backward, Adam(CRNN))
    and Phase B each as one hipGraph replay per shape equal the eager loop: same CRNN and cleaner weights after six steps (the jitter's
    sigma draws and Philox counters follow the same sequence in both runs)."""
    from datasets.synthetic import SyntheticTextAreas
    from train_nn_area import TrainNNPrep
    res = {}
    for flag in (False, True):
        torch.manual_seed(0)
        import random
        random.seed(0)
        tr = SyntheticTextAreas(48, seed=1, include_name=True, include_index=True)
        cers_path = tmp_path / f"cers{int(flag)}.json"
        json.dump({n: (i % 7) / 6 for i, n in enumerate(tr.names)}, open(cers_path, "w"))
        args = _args("a", tmp_path / f"exp{int(flag)}", batch_size=8, inner_limit=2, minibatch_subset="topKCER", minibatch_subset_prop=0.5,
                     cers_ocr_path=str(cers_path), graph=flag)
        t = TrainNNPrep(args, train_set=tr, val_set=SyntheticTextAreas(8, seed=2, include_name=True))
        t.train()
        if flag:
            assert len(t.phase_a_graphs.graphs) >= 1 and len(t.phase_b_graphs.graphs) >= 1
        res[flag] = [torch.cat([p.detach().flatten().clone() for p in m.parameters()]) for m in (t.crnn_model, t.prep_model)]
    for a, b in zip(res[True], res[False]):
        d = (a - b).abs().max().item()
        assert d <= 5e-6, d


def test_area_trainer_graph_replays_phase_b(tmp_path):
    """[new] --graph: Phase B of the area trainer as one hipGraph per shape (two eager steps, then capture, then replays) is the
    same training as the eager loop: same losses step by step, same weights after six steps."""
    from datasets.synthetic import SyntheticTextAreas
    from train_nn_area import TrainNNPrep
    res = {}
    for flag in (False, True):
        torch.manual_seed(0)
        tr = SyntheticTextAreas(48, seed=1, include_name=True, include_index=True)
        args = _args("a", tmp_path / f"exp{int(flag)}", batch_size=8, inner_limit=0, graph=flag)
        t = TrainNNPrep(args, train_set=tr, val_set=SyntheticTextAreas(8, seed=2, include_name=True))
        assert (t.phase_b_graphs is not None) == flag
        losses = []
        if flag:
            orig = t.phase_b_graphs.step
            def spy(X, labels, orig=orig, losses=losses):
                r = orig(X, labels)
                losses.append(None if r is None else float(r[0].item()))
                return r
            t.phase_b_graphs.step = spy
        t.train()
        res[flag] = (torch.cat([p.detach().flatten().clone() for p in t.prep_model.parameters()]), losses)
        if flag:
            assert len(losses) == 6 and losses[0] is None and losses[1] is None and all(l is not None and l == l for l in losses[2:])
            assert len(t.phase_b_graphs.graphs) >= 1
    d = (res[True][0] - res[False][0]).abs().max().item()
    assert d <= 2e-6, d                                        # Adam moves a weight by ~lr = 5e-5 per step


@pytest.mark.parametrize("method", ["levenshtein", "self_attention"])
def test_area_trainer_label_history_weightgens(tmp_path, method):
    """--inner_limit_skip with the non-decaying weight generators: sample-wise CTC (reduction='none') on the HIP path
    (the reference's patch trainer lacks primary_loss_fn_sample_wise; both trainers have it here)."""
    from datasets.synthetic import SyntheticTextAreas
    from train_nn_area import TrainNNPrep
    tr = SyntheticTextAreas(16, seed=3, include_name=True, include_index=True)
    cers_path = tmp_path / "cers.json"
    json.dump({n: 0.5 for n in tr.names}, open(cers_path, "w"))
    args = _args("a", tmp_path / "exp", batch_size=8, minibatch_subset="topKCER", minibatch_subset_prop=0.5, cers_ocr_path=str(cers_path),
                 inner_limit=2, inner_limit_skip=True, window_size=3, weightgen_method=method, epoch=2)
    t = TrainNNPrep(args, train_set=tr, val_set=SyntheticTextAreas(8, seed=4, include_name=True))
    t.train()
    flat = torch.cat([p.detach().flatten() for p in t.crnn_model.parameters()])
    assert torch.isfinite(flat).all()
    assert max(len(v) for v in t.tracked_labels.values()) == 2          # two epochs of history for the selected strips


def test_area_trainer_width_buckets(tmp_path):
    """variable-width lines, one width bucket per batch (BASELINE configs[4] / SURVEY F8 extension)."""
    from datasets.bucketing import bucket_of
    from datasets.synthetic import SyntheticTextAreas
    from train_nn_area import TrainNNPrep
    widths = [96, 128, 160, 240, 256, 300, 400, 512] * 4
    tr = SyntheticTextAreas(32, seed=5, include_name=True, include_index=True, widths=[(w + 15) // 16 * 16 for w in widths])
    args = _args("a", tmp_path / "exp", batch_size=4, inner_limit=1)
    t = TrainNNPrep(args, train_set=tr, val_set=SyntheticTextAreas(8, seed=6, include_name=True))
    seen = set()
    for images, labels, names, idx in t.loader_train:
        assert images.shape[0] == 4 and images.shape[-1] in (128, 256, 384, 512)
        assert all(bucket_of(tr.widths[i]) == images.shape[-1] for i in idx.tolist())
        seen.add(images.shape[-1])
    assert seen == {128, 256, 384, 512}
    t.train()
    assert torch.isfinite(torch.cat([p.detach().flatten() for p in t.prep_model.parameters()])).all()


def test_hipgraph_replay_of_a_phase_b_step_equals_eager():
    """A whole Phase-B step (UNet+CRNN fwd, CTC+MSE, backward incl. the wgrad side stream, capturable Adam) recorded into a
    hipGraph and replayed gives bit-identical weights to the same steps launched eagerly."""
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea.graph import GraphedStep
    from qea.loss import CTCLoss
    from qea.optim import FusedAdam
    B = 8
    x = H.synth_images(B, 5).cuda()
    labels = H.synth_labels(B, 6, 1, 9)
    y, ysz = H.encode(labels)
    y_d, ysz_d = y.cuda(), ysz.cuda()
    ins_d = torch.full((B,), 31, dtype=torch.int32, device="cuda")
    ones = torch.ones(B, 1, 32, 128, device="cuda")

    def build():
        prep = UNet()
        prep.load_state_dict(mo.seeded_state(mo.unet_state_shapes(), 3))
        crnn = CRNN(95, False)
        crnn.load_state_dict(mo.seeded_state(mo.crnn_state_shapes(), 4))
        prep, crnn = prep.cuda(), crnn.cuda()
        crnn.register_backward_hook(crnn.backward_hook)
        opt = FusedAdam(prep.parameters(), lr=5e-4, capturable=True)
        ctc = CTCLoss()
        ctc.max_target_length = int(ysz.max())
        prep.train()
        crnn.train()
        for m in crnn.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()

        def step():
            prep.zero_grad()
            crnn.zero_grad()
            img = prep(x)
            loss = ctc(crnn(img), y_d, ins_d, ysz_d) + torch.nn.functional.mse_loss(img, ones)
            loss.backward()
            opt.step()
            return loss
        return prep, step

    n_warm, n_replay = 2, 3
    prep_e, step_e = build()
    for _ in range(n_warm + n_replay):
        loss_e = step_e()
    prep_g, step_g = build()
    g = GraphedStep(step_g, warmup=n_warm)
    for _ in range(n_replay):
        loss_g = g()
    torch.cuda.synchronize()
    assert torch.isfinite(loss_g).all() and torch.equal(loss_g, loss_e)
    for (k, a), (_, b) in zip(prep_e.state_dict().items(), prep_g.state_dict().items()):
        assert torch.equal(a, b), k


def test_eager_forward_after_graph_replays_sees_the_new_weights():
    """a replayed hipGraph updates the weights through raw pointers and runs no Python, so nothing moved the
    weight-cache key (qea.ops.weight_cached) — an eager forward AFTER replays used the derived filter planes of the weights
    BEFORE them.  Sequence: eager eval forward (cache filled) -> replays -> eager eval forward, against the same forward with
    the cache off."""
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea import ops
    from qea.graph import GraphedStep
    from qea.optim import FusedAdam
    B = 8
    x = H.synth_images(B, 15).cuda()
    ones = torch.ones(B, 1, 32, 128, device="cuda")
    prep = UNet()
    prep.load_state_dict(mo.seeded_state(mo.unet_state_shapes(), 3))
    prep = prep.cuda()
    opt = FusedAdam(prep.parameters(), lr=1e-3, capturable=True)

    def step():
        prep.train()
        prep.zero_grad()
        loss = torch.nn.functional.mse_loss(prep(x), ones)
        loss.backward()
        opt.step()
        return loss

    def eval_forward():
        prep.eval()
        with torch.no_grad():
            return prep(x).clone()
    g = GraphedStep(step, warmup=1)
    before = eval_forward()                                   # fills the cache with the forms of the CURRENT weights
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    cached = eval_forward()
    ops.WEIGHT_CACHE["on"] = False
    try:
        fresh = eval_forward()
    finally:
        ops.WEIGHT_CACHE["on"] = True
    assert not torch.equal(before, fresh)                     # the replays did move the weights
    assert torch.equal(cached, fresh)


def _dp_gpu_worker(rank, world, port, tmp):
    """two ranks SHARING cuda:0 (gloo moves the flat gradient buffers through the host): the product's HIP backend end to end under
    the data-parallel path — equal shards, whole-minibatch TopKCER over the ranks, winner re-balance, two all-reduces per step"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      QEA_DIST_BACKEND="gloo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "query-efficient-approx-to-improve-ocr_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from datasets.synthetic import SyntheticTextAreas
    from qea import dist as qdist
    from train_nn_area import TrainNNPrep
    tr = SyntheticTextAreas(16, seed=3, include_name=True, include_index=True)
    cers_path = os.path.join(tmp, f"cers{rank}.json")
    json.dump({n: ((i * 7) % 16) / 16 for i, n in enumerate(tr.names)}, open(cers_path, "w"))
    args = _args("a", os.path.join(tmp, f"exp{rank}"), batch_size=4, minibatch_subset="topKCER", minibatch_subset_prop=0.5,
                 cers_ocr_path=cers_path, inner_limit=2)
    t = TrainNNPrep(args, train_set=tr, val_set=SyntheticTextAreas(4, seed=4, include_name=True))
    assert t.world == world and t.device.type == "cuda"
    seen = []
    rb = qdist.rebalance_rows
    qdist.rebalance_rows = lambda rows, counts: (seen.append((rows.shape[0], list(counts))), rb(rows, counts))[1]
    t.train()
    flat = torch.cat([p.detach().flatten() for p in t.prep_model.parameters()] + [p.detach().flatten() for p in t.crnn_model.parameters()]).cpu()
    picked = sorted(n for n, v in t.selected_samples.items() if v[0])
    torch.save({"flat": flat, "picked": picked, "rebalanced": seen, "steps": len(t.loader_train)}, os.path.join(tmp, f"dp_r{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_two_ranks_on_one_gpu(tmp_path):
    """SURVEY §8e on hardware, as far as one card allows: 2 ranks x batch 4 on the HIP backend (gloo in place of RCCL, which refuses two
    ranks on one device): both ranks run the same number of steps, hold bit-identical weights after every pair of all-reduces, the
    union of their picks is the global top-k of each minibatch and the winners were dealt out in equal slices."""
    import torch.multiprocessing as mp
    port = 32500 + os.getpid() % 2000
    mp.start_processes(_dp_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "dp_r0.pt"), torch.load(tmp_path / "dp_r1.pt")
    assert r0["steps"] == r1["steps"] == 2                                        # 16 samples / (2 ranks x 4)
    assert torch.equal(r0["flat"], r1["flat"]) and torch.isfinite(r0["flat"]).all()
    assert len(r0["picked"]) + len(r1["picked"]) == 2 * 4 and not set(r0["picked"]) & set(r1["picked"])   # k = 4 of each global batch of 8
    for a, b in zip(r0["rebalanced"], r1["rebalanced"]):
        assert a[1] == b[1] and sum(a[1]) == 4 and a[0] == a[1][0] and b[0] == b[1][1]                     # the same plan on both ranks
