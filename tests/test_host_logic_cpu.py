"""Host-side mirror of the reference interface (samplers, helpers, CLI flags, constants) on CPU."""
import json
import os

import numpy as np
import torch

import helpers as H


def test_properties_constants():
    import properties
    assert properties.char_set == H.CHAR_SET and properties.char_set[0] == "`"
    assert properties.input_size == (32, 128) and properties.max_char_len == 100 and properties.empty_char == " "
    assert properties.prep_crnn_ckpts == "ckpts" and properties.img_out == "img_out" and properties.param_path == "params.txt"
    assert properties.patch_dataset_train == "patch_dataset_train" and properties.vgg_text_dataset_dev == "vgg_dev"


def test_topk_sampler_matches_reference_cases(golden_dir):
    from selection_utils import datasampler_factory
    cases = json.load(open(os.path.join(golden_dir, "topk_cases.json")))["cases"]
    for c in cases:
        s = datasampler_factory("topKCER")(dict(c["cers"]))
        imgs = torch.arange(len(c["names"])).float().view(-1, 1)
        sel, labels, idx = s.query(imgs, list(c["names"]), c["k"], list(c["names"]))
        vals = np.float32([c["cers"][n] for n in c["names"] if n in c["cers"]])
        if "build-specific" in c["note"] or "real slice" in c["note"]:
            assert sorted(vals[idx].tolist()) == sorted(vals[c["idx"]].tolist())
        else:
            assert idx.tolist() == c["idx"], c["note"]
            if c["sel"] is not None:
                assert sel.view(-1).long().tolist() == c["sel"]
        assert labels == [c["names"][i] for i in idx.tolist()]


def test_sampler_protocol_and_factory_keys():
    from selection_utils import datasampler_factory
    for key in ("random", "topKCER", "uniformCERglobal", "randomglobal", "rangeCER", "uniformEntropy"):
        assert datasampler_factory(key)
    s = datasampler_factory("topKCER")({"a": 0.1})
    s.update_cer([0.5, 0.25], ["a", "b"])
    assert s.cers == {"a": 0.5, "b": 0.25} and s.all_cers == {"a": [0.5], "b": [0.25]}
    imgs = torch.arange(6).float().view(-1, 1)
    r = datasampler_factory("random")()
    sel, lab, idx = r.query(imgs, list("abcdef"), 3)
    assert sel.shape[0] == 3 and len(set(idx.tolist())) == 3
    rg = datasampler_factory("rangeCER")({n: i / 10 for i, n in enumerate("abcdef")})
    sel, lab, idx = rg.query(imgs, list("abcdef"), 2, list("abcdef"))
    assert len(set(idx.tolist())) == 2
    g = datasampler_factory("uniformCERglobal")({n: i / 10 for i, n in enumerate("abcdef")}, 2)
    g.select_samples()
    sel, lab, idx = g.query(imgs, list("abcdef"), names=list("abcdef"))
    assert len(idx) == 2


def test_utils_helpers_match_reference_fixtures(tmp_path):
    import utils
    fx = H.golden("helpers.npz")
    c2i, i2c, n = utils.get_char_maps(H.CHAR_SET)
    assert n == 95 and c2i["`"] == 0 and i2c[94] == "/"
    assert utils.pred_to_string(torch.from_numpy(fx["dec|scores"]), [""] * 6, i2c) == [str(s) for s in fx["dec|strings"]]
    boxes = [dict(label=str(i), x_min=int(b[0]), y_min=int(b[1]), x_max=int(b[2]), y_max=int(b[3])) for i, b in enumerate(fx["crop|boxes"])]
    stack, labels = utils.get_text_stack(torch.from_numpy(fx["crop|page"]), boxes, (32, 128))
    assert np.array_equal(stack.numpy(), fx["crop|stack"]) and labels == ["0", "1", "2", "3"]
    assert utils.compare_labels(["kitten", "abc"], ["sitting", "abc"]) == (1, 3 / 7)
    bn = torch.nn.BatchNorm2d(3).train()
    utils.set_bn_eval(bn)
    assert not bn.training

    class T:
        pass
    t, args = T(), type("A", (), dict(crnn_model=None, prep_model=None, data_base_path=".", exp_base_path=str(tmp_path / "exp")))()
    utils.create_dirs(t, args)
    for d in ("ckpts", "img_out", "cers", "tracked_labels", "selected_samples"):
        assert os.path.isdir(tmp_path / "exp" / d)
    utils.save_img(torch.rand(3, 1, 8, 16), "grid", str(tmp_path), 2)
    assert os.path.exists(tmp_path / "grid.png")


def test_add_gaussian_noice_call_semantics():
    from oracle import path_oracle as po
    from transform_helper import AddGaussianNoice
    fx = H.golden("helpers.npz")
    img = torch.from_numpy(fx["jit|img"])
    for stochastic, tag in ((False, "jit0"), (True, "jit1")):
        torch.manual_seed(123)
        out, noise = AddGaussianNoice(std=5, is_stochastic=stochastic, return_noise=True)(img.clone())
        assert np.array_equal(noise.numpy(), fx[f"{tag}|noise"]) and np.array_equal(out.numpy(), fx[f"{tag}|out"])
    torch.manual_seed(123)
    out = AddGaussianNoice(std=3)(img.clone(), noise_coef=0.5)
    assert np.array_equal(out.numpy(), po.jitter(img.numpy(), fx["jitc|noise"], 0.5))


def test_tracking_utils_and_stub_ocr():
    import tracking_utils as tu
    from label_tracking.tracking_methods import weightgenerator_factory
    from ocr_helper.stub_helper import StubHelper

    class T:
        window_size, weightgen_method, char_to_index = 3, "decaying", H.C2I
        tracked_labels = {"a": ["x", "yy"], "b": ["z"], "c": []}
    t = T()
    tu.add_labels_to_history(t, ["a", "c"], ["new", "cc"])
    assert t.tracked_labels["a"][-1] == "new" and t.tracked_labels["c"] == ["cc"]
    batches = tu.generate_ctc_target_batches(t, ["a", "b", "c"])
    assert [b[2] for b in batches] == [[0, 1, 2], [0], [0]]
    assert batches[0][1].tolist() == [3, 1, 2]
    args = type("A", (), dict(decay_factor=0.5, window_size=3))()
    w = weightgenerator_factory("decaying")(args, "cpu").gen_weights(None, None)
    assert w.tolist() == [1.0, 0.5, 0.25]
    ocr = StubHelper()
    imgs = H.synth_images(5, 3)
    a, b = ocr.get_labels(imgs), ocr.get_labels(imgs)
    assert a == b and ocr.count_calls == 10 and all(set(l) <= set(H.CHAR_SET) and 0 < len(l) <= 100 for l in a)


def test_cli_flag_surface():
    from qea.cli_flags import build_parser
    p = build_parser("p", "").parse_args([])
    assert (p.lr_crnn, p.lr_prep, p.epoch, p.std, p.inner_limit, p.weight_decay, p.minibatch_subset_prop) == (1e-4, 5e-5, 25, 5, 2, 5e-4, 0.5)
    assert p.random_std is True and p.ocr == "Tesseract" and p.weightgen_method == "decaying" and p.decay_factor == 0.7
    a = build_parser("a", "").parse_args(["--minibatch_subset", "topKCER", "--random_std", "--lr_scheduler", "cosine"])
    assert a.batch_size == 32 and a.epoch == 50 and a.random_std is False and not hasattr(a, "weight_decay")


def test_file_datasets_roundtrip(tmp_path):
    """on-disk formats of the reference's datasets (datasets/img_dataset.py, patch_dataset.py)."""
    from PIL import Image
    from datasets._io import to_tensor
    from datasets.img_dataset import ImgDataset
    from datasets.patch_dataset import PatchDataset
    from transform_helper import PadWhite
    d = tmp_path / "vgg_train"
    d.mkdir()
    rng = np.random.RandomState(0)
    for i, lab in enumerate(["hello", "A1", "x" * 101]):
        Image.fromarray((rng.rand(20, 60) * 255).astype(np.uint8)).save(d / f"{i}_{lab}_w.png")
    ds = ImgDataset(str(d), transform=lambda im: to_tensor(PadWhite((32, 128))(im)), include_name=True, include_index=True)
    assert len(ds) == 2                                   # the 101-char label is dropped
    img, label, name, idx = ds[0]
    assert img.shape == (1, 32, 128) and label in ("hello", "A1") and name.endswith("_w.png") and idx == 0
    assert img[0, 0, 0] == 1.0 and img.min() < 1.0        # white padding around the strip
    p = tmp_path / "patch_dataset_train" / "folderA"
    p.mkdir(parents=True)
    Image.fromarray((rng.rand(100, 300) * 255).astype(np.uint8)).save(p / "doc1.png")
    boxes = [dict(label="ab", x_min=10, y_min=5, x_max=90, y_max=30), dict(label="wide", x_min=0, y_min=0, x_max=200, y_max=20)]
    json.dump(boxes, open(p / "doc1.json", "w"))
    pd = PatchDataset(str(tmp_path / "patch_dataset_train"), pad=True, include_name=True)
    image, labels, name = pd[0]
    assert image.shape == (1, 400, 512) and len(labels) == 1 and labels[0]["label"] == "ab"
    assert (labels[0]["x_min"], labels[0]["y_min"]) == (10 + 106, 5 + 150)      # centred on the 400x512 canvas
    batch = PatchDataset.collate([pd[0]])
    assert batch[0].shape == (1, 1, 400, 512) and batch[2][0].endswith("doc1.png")


def test_levenshtein_and_attention_weight_generators():
    from label_tracking.tracking_methods import weightgenerator_factory
    args = type("A", (), dict(decay_factor=0.5, window_size=3, query_dim=8, emb_dim=16, attn_activation="sigmoid"))()
    hist = {"a": ["cat", "cat", "cut", "cat"], "b": ["x"], "c": []}
    lev = weightgenerator_factory("levenshtein")(args, "cpu")
    w = lev.gen_weights(hist, ["a", "b", "c", "missing"])
    assert w.shape == (4, 4) and w[:, 0].tolist() == [1, 1, 1, 1]
    # "a": window = [cat, cut, cat] (most recent first); cat vs {cut, cat}: mean 0.5 -> 0.5*(1-0.5/3)
    assert abs(w[0, 1].item() - 0.5 * (1 - 0.5 / 3)) < 1e-6 and abs(w[0, 2].item() - 0.5 * (1 - 1 / 3)) < 1e-6
    assert abs(w[1, 1].item() - 0.5) < 1e-6 and w[2, 1:].abs().sum() == 0 and w[3, 1:].abs().sum() == 0
    torch.manual_seed(0)
    att = weightgenerator_factory("self_attention")(args, "cpu", H.C2I)
    wa = att.gen_weights(hist, ["a", "b", "c"])
    assert wa.shape == (3, 4) and (wa[0, 1:] > 0).all() and (wa[0, 1:] < 1).all() and wa[1, 2:].abs().sum() == 0
    assert set(att.attention_model.state_dict().keys()) == {"positional_encodings", "embedding", "Wq.weight", "Wq.bias",
                                                            "loss_coef_layer.weight", "loss_coef_layer.bias"}


def test_global_topk_reports_the_picked_count_when_the_minibatch_is_smaller_than_k():
    """with fewer rows than k_global every row is picked and k is that number on every path (share = world * n / k)."""
    from qea import dist as qdist
    order, k = qdist.global_topk([0.3, 0.9], 5)
    assert order.tolist() == [1, 0] and k == 2
    order, k, counts = qdist.global_topk([0.3, 0.9], 5, with_counts=True)
    assert k == 2 and counts == [2] and k == sum(counts)
    order, k = qdist.global_topk([0.1, 0.5, 0.4], 2)
    assert order.tolist() == [1, 2] and k == 2
