"""Plumbing of the two trainers on CPU (BASELINE configs[0]): the product's host logic (flags ->
TrainNNPrep -> Phase A/B loop -> experiment files and whole-module checkpoints) driven with the CPU
oracle INJECTED as the arithmetic backend, the stub OCR and synthetic data; plus the data-parallel
path with gloo at world_size 2 against its single-process specification."""
import json
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

torch.set_num_threads(4)


def oracle_backend():
    from oracle.modules import OracleCRNN, OracleUNet
    from qea.trainer_core import Backend
    return Backend(OracleUNet, OracleCRNN, torch.nn.CTCLoss, torch.optim.Adam, torch.device("cpu"), gpu_jitter=False)


def _args(which, tmp, **over):
    from qea.cli_flags import build_parser
    a = build_parser(which, "").parse_args(["--exp_base_path", str(tmp), "--ocr", "stub", "--epoch", "1", "--inner_limit", "1"])
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_area_trainer_plumbing(tmp_path):
    from datasets.synthetic import SyntheticTextAreas
    from ocr_helper.stub_helper import StubHelper
    from train_nn_area import TrainNNPrep
    tr_set = SyntheticTextAreas(8, seed=1, include_name=True, include_index=True)
    cers = {n: float(i % 3) / 2 for i, n in enumerate(tr_set.names)}
    cers_path = tmp_path / "cers.json"
    json.dump(cers, open(cers_path, "w"))
    args = _args("a", tmp_path / "exp", batch_size=4, minibatch_subset="topKCER", minibatch_subset_prop=0.5, cers_ocr_path=str(cers_path),
                 inner_limit=2, inner_limit_skip=True, window_size=2)
    ocr = StubHelper()
    t = TrainNNPrep(args, backend=oracle_backend(), train_set=tr_set, val_set=SyntheticTextAreas(4, seed=2, include_name=True), ocr=ocr)
    w0 = [p.detach().clone() for p in t.prep_model.parameters()]
    c0 = [p.detach().clone() for p in t.crnn_model.parameters()]
    best = t.train()
    assert isinstance(best, tuple) and len(best) == 2
    assert any((a != b).any() for a, b in zip(w0, t.prep_model.parameters()))        # Phase B stepped the UNet
    assert any((a != b).any() for a, b in zip(c0, t.crnn_model.parameters()))        # Phase A stepped the CRNN
    # 2 minibatches x k=2 strips x 2 inner iterations (+ validation: 4) black-box calls
    assert ocr.count_calls == 2 * 2 * 2 + 4
    exp = tmp_path / "exp"
    assert os.path.exists(exp / "ckpts" / "CRNN_model_0") and any(f.startswith("Prep_model_0_") for f in os.listdir(exp / "ckpts"))
    assert json.load(open(exp / "cers" / "all_cers.json")).keys() == cers.keys()
    tracked = json.load(open(exp / "tracked_labels" / "tracked_labels_current.json"))
    assert sum(len(v) for v in tracked.values()) == 4                                # one history entry per selected strip
    sel = json.load(open(exp / "selected_samples" / "selected_samples_current.json"))
    assert sum(v[0] for v in sel.values()) == 4
    assert os.path.exists(exp / "img_out" / "out_0.png")
    # the sampler's CERs were refreshed from the CRNN's greedy decodes
    assert set(t.sampler.all_cers.keys()) == set(tr_set.names)
    crnn = torch.load(exp / "ckpts" / "CRNN_model_0", weights_only=False)
    assert list(crnn.state_dict().keys())


def test_area_trainer_select_before_clean_is_the_same_training(tmp_path):
    """[new] --select_before_clean: TopKCER ranks names / CERs only, so cleaning just the picked images (eval-mode UNet) must
    train the very same CRNN as cleaning the whole minibatch and then picking (train_nn_area.py:217-225)."""
    from datasets.synthetic import SyntheticTextAreas
    from ocr_helper.stub_helper import StubHelper
    from train_nn_area import TrainNNPrep
    outs = []
    for flag in (False, True):
        tr_set = SyntheticTextAreas(8, seed=1, include_name=True, include_index=True)
        cers_path = tmp_path / f"cers{int(flag)}.json"
        json.dump({n: float(i % 5) / 4 + 0.01 * i for i, n in enumerate(tr_set.names)}, open(cers_path, "w"))
        args = _args("a", tmp_path / f"exp{int(flag)}", batch_size=4, minibatch_subset="topKCER", minibatch_subset_prop=0.5,
                     cers_ocr_path=str(cers_path), inner_limit=2, select_before_clean=flag)
        t = TrainNNPrep(args, backend=oracle_backend(), train_set=tr_set, val_set=SyntheticTextAreas(4, seed=2, include_name=True), ocr=StubHelper())
        t.train()
        outs.append((torch.cat([p.detach().flatten() for p in t.crnn_model.parameters()]),
                     torch.cat([p.detach().flatten() for p in t.prep_model.parameters()]),
                     sorted(n for n, v in t.selected_samples.items() if v[0])))
    assert outs[0][2] == outs[1][2]
    assert torch.allclose(outs[0][0], outs[1][0], rtol=0, atol=2e-4) and torch.allclose(outs[0][1], outs[1][1], rtol=0, atol=1e-4)


def test_patch_trainer_plumbing(tmp_path):
    from datasets.synthetic import SyntheticPatches
    from ocr_helper.stub_helper import StubHelper
    from train_nn_patch import TrainNNPrep
    tr_set = SyntheticPatches(2, seed=1, strips=(2, 3), pad_shape=(80, 256))
    names = []
    for i in range(len(tr_set)):
        _, boxes, name = tr_set[i]
        names += TrainNNPrep._strip_names([b["label"] for b in boxes], name)
    cers_path = tmp_path / "cers.json"
    json.dump({n: 0.5 for n in names}, open(cers_path, "w"))
    args = _args("p", tmp_path / "exp", minibatch_subset="topKCER", minibatch_subset_prop=0.5, cers_ocr_path=str(cers_path), inner_limit=2,
                 update_CRNN=True)
    ocr = StubHelper()
    t = TrainNNPrep(args, backend=oracle_backend(), train_set=tr_set, val_set=SyntheticPatches(1, seed=2, strips=(2, 2), pad_shape=(80, 256),
                                                                                             include_name=False), ocr=ocr)
    t.train()
    exp = tmp_path / "exp"
    for f in ("CRNN_model_0", "optim_prep_latest", "optim_crnn_latest"):
        assert os.path.exists(exp / "ckpts" / f), f
    st = torch.load(exp / "ckpts" / "optim_prep_latest", weights_only=False)
    assert set(st.keys()) == {"state", "param_groups"} and st["param_groups"][0]["weight_decay"] == 5e-4
    assert set(json.load(open(exp / "cers" / "all_cers.json")).keys()) == set(names)
    assert ocr.count_calls > 0


def test_patch_trainer_docs_per_step_equals_the_sequential_loop(tmp_path, monkeypatch):
    """[new] --docs_per_step N (SURVEY §8 f2 "multi-document batching"): Phase B runs the N documents through the cleaner as one
    batch with per-document BatchNorm statistics and all their strips through the CRNN as one batch, sums the N per-document
    losses and back-propagates once.  Against the reference's loop written out (train_nn_patch.py:318-329: per document forward,
    loss, backward — gradients accumulate) on the same models: the gradient Adam(UNet) consumes and the UNet's running statistics."""
    import torch.nn.functional as F
    from datasets.synthetic import SyntheticPatches
    from ocr_helper.stub_helper import StubHelper
    from oracle.modules import OracleCRNN, OracleUNet
    from train_nn_patch import TrainNNPrep
    from utils import get_text_stack
    tr_set = SyntheticPatches(4, seed=1, strips=(2, 3), pad_shape=(80, 256))
    args = _args("p", tmp_path / "exp", inner_limit=0, docs_per_step=2)
    t = TrainNNPrep(args, backend=oracle_backend(), train_set=tr_set, val_set=SyntheticPatches(1, seed=2, strips=(2, 2), pad_shape=(80, 256),
                                                                                             include_name=False), ocr=StubHelper())
    assert len(t.loader_train) == 2                                              # 4 documents, 2 per step
    seen, grads = [], []
    fwd = type(t.prep_model).forward                                             # patched on the class: the trainer pickles the module

    def spy_fwd(self, x, bn_groups=1):
        if torch.is_grad_enabled() and self is t.prep_model:
            seen.append((x.detach().clone(), bn_groups))
        return fwd(self, x, bn_groups=bn_groups)
    monkeypatch.setattr(type(t.prep_model), "forward", spy_fwd)
    step = t._step_prep

    def spy_step(also_crnn=False):
        if not grads:
            grads.append(({n: p.grad.detach().clone() for n, p in t.prep_model.named_parameters()},
                          {n: b.detach().clone() for n, b in t.prep_model.named_buffers()}))
        step(also_crnn)
    t._step_prep = spy_step
    batches = []
    coll = t._collate

    def spy_collate(items):
        batches.append(items)
        return coll(items)
    t.loader_train = torch.utils.data.DataLoader(tr_set, batch_size=2, shuffle=False, collate_fn=spy_collate)
    t.train()
    assert seen[0][1] == 2 and seen[0][0].shape[0] == 2
    # ---- the reference's loop on identically initialised models
    prep, crnn = OracleUNet(), OracleCRNN(t.vocab_size)
    crnn.register_backward_hook(crnn.backward_hook)
    prep.train(); crnn.train(); crnn.apply(__import__("utils").set_bn_eval)
    for image, boxes, _name in batches[0]:
        img_out = prep(image.unsqueeze(0))[0]
        crops, labels = get_text_stack(img_out, boxes, (32, 128))
        scores = crnn(crops)
        y, ysz = H.encode(labels)
        loss = F.ctc_loss(scores, y, torch.full((len(labels),), scores.shape[0], dtype=torch.int), ysz) + \
            F.mse_loss(img_out, torch.ones_like(img_out)) * args.scalar
        loss.backward()
    g_ref = dict(prep.named_parameters())
    for n, g in grads[0][0].items():
        ref = g_ref[n].grad
        assert (g - ref).norm().item() <= 2e-5 * max(ref.norm().item(), 1e-12), n
    for n, b in prep.named_buffers():
        assert torch.allclose(grads[0][1][n].double(), b.double(), rtol=1e-6, atol=1e-7), n


# ----------------------------------------------------------------------------- data parallel (gloo, 2 ranks)
def _dp_worker(rank, world, port, tmp, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "query-efficient-approx-to-improve-ocr_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    from datasets.synthetic import SyntheticTextAreas
    from ocr_helper.stub_helper import StubHelper
    from qea import dist as qdist
    from train_nn_area import TrainNNPrep
    # 7 samples on 2 ranks x batch 2: NOT divisible — every rank must still run the same number of steps;
    # TopKCER with prop 0.5 picks k = 2 of each GLOBAL batch of 4 (train_nn_area.py:220-225 ranks the whole minibatch)
    tr_set = SyntheticTextAreas(7, seed=1, include_name=True, include_index=True)
    cers = {n: c for n, c in zip(tr_set.names, [0.30, 0.90, 0.10, 0.70, 0.20, 0.80, 0.40])}
    cers_path = os.path.join(tmp, f"cers{rank}.json")
    json.dump(cers, open(cers_path, "w"))
    args = _args("a", os.path.join(tmp, f"exp{rank}"), batch_size=2, inner_limit=1, minibatch_subset="topKCER", minibatch_subset_prop=0.5,
                 cers_ocr_path=cers_path)
    t = TrainNNPrep(args, backend=oracle_backend(), train_set=tr_set, val_set=SyntheticTextAreas(2, seed=2, include_name=True), ocr=StubHelper())
    assert t.world == world
    shard = sorted(tr_set.names[i] for batch in t.loader_train.sampler for i in [batch])
    assert len(t.loader_train) == 1 and len(shard) == 2
    # gradient exchange == mean of the shard gradients (checked by the parent): one Phase-B backward on a rank-specific shard
    x = H.synth_images(2, 100 + rank)
    t._set_phase_b()
    img = t.prep_model(x)
    scores, y, ps, ys = t._call_model(img, H.synth_labels(2, 200 + rank, 2, 6))
    t._get_loss(scores, y, ps, ys, img).backward()
    local = torch.cat([p.grad.flatten().clone() for p in t.prep_model.parameters()])
    qdist.allreduce_module_grads(t.prep_model)
    reduced = torch.cat([p.grad.flatten().clone() for p in t.prep_model.parameters()])
    t.train()
    mine, k = qdist.global_topk([0.1 * (rank + 1), 0.9 - 0.5 * rank, 0.3], 3)
    flat = torch.cat([p.detach().flatten() for p in t.prep_model.parameters()] + [p.detach().flatten() for p in t.crnn_model.parameters()])
    picked = sorted(n for n, v in t.selected_samples.items() if v[0])
    torch.save({"flat": flat, "topk": mine, "local": local, "reduced": reduced, "shard": shard, "picked": picked, "cers0": cers},
               os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_gloo_world2(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.start_processes(_dp_worker, args=(2, port, str(tmp_path), str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat"], r1["flat"])                 # ranks stay in lock-step after A and B updates
    # global TopK over the sharded CERs [0.1,0.9,0.3 | 0.2,0.4,0.3], k=3 -> 0.9 (rank0 idx1), 0.4 (rank1 idx1), 0.3 (rank0 idx2: ties rank-major)
    assert r0["topk"].tolist() == [1, 2] and r1["topk"].tolist() == [1]

    # specification: the exchanged gradient is the mean of the per-shard gradients, identical on all ranks
    assert torch.equal(r0["reduced"], r1["reduced"])
    assert torch.allclose(r0["reduced"], (r0["local"] + r1["local"]) / 2, rtol=1e-6, atol=1e-9)
    assert not torch.equal(r0["local"], r1["local"])
    # whole-minibatch selection under DP == the single-process selection over the concatenated minibatch
    assert not set(r0["shard"]) & set(r1["shard"])
    batch = r0["shard"] + r1["shard"]
    want = sorted(sorted(batch, key=lambda n: -r0["cers0"][n])[:2])
    assert sorted(r0["picked"] + r1["picked"]) == want, (r0["picked"], r1["picked"], want)
    assert set(r0["picked"]) <= set(r0["shard"]) and set(r1["picked"]) <= set(r1["shard"])


def _winners_setup(tmp, tag, batch_size, n=4, **over):
    from datasets.synthetic import SyntheticTextAreas
    from ocr_helper.stub_helper import StubHelper
    from train_nn_area import TrainNNPrep
    tr_set = SyntheticTextAreas(n, seed=1, include_name=True, include_index=True)
    cers_path = os.path.join(tmp, f"cers_{tag}.json")
    json.dump({nm: 0.5 for nm in tr_set.names}, open(cers_path, "w"))
    args = _args("a", os.path.join(tmp, f"exp_{tag}"), batch_size=batch_size, inner_limit=2, std=0, minibatch_subset="topKCER",
                 minibatch_subset_prop=0.5, cers_ocr_path=cers_path, **over)
    t = TrainNNPrep(args, backend=oracle_backend(), train_set=tr_set, val_set=SyntheticTextAreas(batch_size, seed=2, include_name=True), ocr=StubHelper())
    return t, tr_set


def _spy_first_crnn_step(t, rec):
    """records what the first Adam(CRNN) step consumes — the (all-reduced) gradient; the parameters after the step are not
    comparable across runs element by element: Adam's first step is +-lr on the SIGN of every gradient element, rounding
    noise included"""
    orig = t.optimizer_crnn.step

    def spy(*a, **kw):
        if "crnn_grad_A" not in rec:
            rec["crnn_grad_A"] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().flatten().clone()
                                            for p in t.crnn_model.parameters()])
        return orig(*a, **kw)
    t.optimizer_crnn.step = spy
    rl = t._replica_losses

    def spy_rl(imgs, noiser, R, **kw):
        rec.setdefault("phase_a_images", []).append(imgs.detach().clone())
        return rl(imgs, noiser, R, **kw)
    t._replica_losses = spy_rl


def _dp_winners_worker(rank, world, port, tmp, rebalance):
    """2 ranks x batch 2 = one global minibatch of 4, k = 2, and BOTH global winners live in rank 0's shard."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "query-efficient-approx-to-improve-ocr_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    from qea import dist as qdist
    t, tr_set = _winners_setup(tmp, f"r{rank}_{int(rebalance)}", 2, no_rebalance_topk=not rebalance)
    shard = [tr_set.names[i] for i in t.loader_train.sampler]
    shards = [None] * world
    dist.all_gather_object(shards, shard)
    for r, names in enumerate(shards):                        # the same table on every rank: rank 0's strips carry the two worst CERs
        for j, nm in enumerate(names):
            t.sampler.cers[nm] = (0.9 - 0.1 * j) if r == 0 else (0.2 - 0.1 * j)
    rec = {"shards": shards}
    _spy_first_crnn_step(t, rec)
    t.train()
    rec["flat"] = torch.cat([p.detach().flatten() for p in t.prep_model.parameters()] + [p.detach().flatten() for p in t.crnn_model.parameters()])
    rec["picked"] = sorted(nm for nm, v in t.selected_samples.items() if v[0])
    torch.save(rec, os.path.join(tmp, f"win_r{rank}_{int(rebalance)}.pt"))
    dist.destroy_process_group()


def test_all_global_winners_on_one_rank(tmp_path):
    """VERDICT r2 #8 / SURVEY §8e.  (a) winners processed where they live (--no_rebalance_topk): rank 0 runs the whole of Phase A,
    rank 1 joins the all-reduce with a zero gradient, and the gradient Adam(CRNN) consumes equals the SINGLE-PROCESS gradient on
    the concatenated minibatch.  (b) default: the winners are dealt out again in equal slices — rank 1 processes rank 0's second
    winner, the ranks stay in lock-step, and the bookkeeping (selected_samples) stays with the owner."""
    tmp = str(tmp_path)
    port = 31500 + os.getpid() % 2000
    for rebalance in (False, True):
        mp.start_processes(_dp_winners_worker, args=(2, port + int(rebalance), tmp, rebalance), nprocs=2, join=True, start_method="spawn")
    a0, a1 = torch.load(tmp_path / "win_r0_0.pt"), torch.load(tmp_path / "win_r1_0.pt")
    b0, b1 = torch.load(tmp_path / "win_r0_1.pt"), torch.load(tmp_path / "win_r1_1.pt")
    for r0, r1 in ((a0, a1), (b0, b1)):
        assert torch.equal(r0["flat"], r1["flat"])                               # lock-step after both updates
        assert r0["picked"] == sorted(r0["shards"][0]) and r1["picked"] == []    # both winners are rank 0's; bookkeeping with the owner
    assert [x.shape[0] for x in a0["phase_a_images"]] == [2] and "phase_a_images" not in a1    # (a) rank 1 had no Phase-A work
    assert [x.shape[0] for x in b0["phase_a_images"]] == [1] and [x.shape[0] for x in b1["phase_a_images"]] == [1]   # (b) one winner each
    # (b) what travelled: rank 0's winners in rank-major order = its descending-CER order; rank 1 got the second one, bit for bit
    assert torch.equal(b0["phase_a_images"][0][0], a0["phase_a_images"][0][0]) and torch.equal(b1["phase_a_images"][0][0], a0["phase_a_images"][0][1])
    # (a) == the single-process update: one process, the four strips as ONE minibatch, the same CER table
    t, tr_set = _winners_setup(tmp, "single", 4)
    assert t.world == 1
    for r, names in enumerate(a0["shards"]):
        for j, nm in enumerate(names):
            t.sampler.cers[nm] = (0.9 - 0.1 * j) if r == 0 else (0.2 - 0.1 * j)
    rec = {}
    _spy_first_crnn_step(t, rec)
    t.train()
    assert sorted(nm for nm, v in t.selected_samples.items() if v[0]) == a0["picked"]
    g1, g2 = rec["crnn_grad_A"].double(), a0["crnn_grad_A"].double()
    assert (g1 - g2).norm().item() <= 1e-5 * g1.norm().item(), ((g1 - g2).norm() / g1.norm()).item()
    assert torch.equal(a0["crnn_grad_A"], a1["crnn_grad_A"])                     # the zero-gradient rank received the same mean


def test_equal_shards_give_every_rank_the_same_step_count(monkeypatch):
    """qea.dist.equal_shards / deal_batches: any dataset size, any world size -> identical step counts, disjoint shards."""
    from qea import dist as qdist
    for n, w, bs in ((10, 4, 1), (255, 2, 64), (7, 2, 2), (1000, 8, 32), (3, 4, 1)):
        shards = []
        for r in range(w):
            monkeypatch.setattr(qdist, "world", lambda w=w: w)
            monkeypatch.setattr(qdist, "rank", lambda r=r: r)
            shards.append(qdist.equal_shards(torch.arange(n), bs).tolist())
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == n // (w * bs) * bs
        flat = [i for s in shards for i in s]
        assert len(set(flat)) == len(flat)
        batches = [[i] for i in range(n)]
        dealt = []
        for r in range(w):
            monkeypatch.setattr(qdist, "rank", lambda r=r: r)
            dealt.append(qdist.deal_batches(batches))
        assert len({len(d) for d in dealt}) == 1
