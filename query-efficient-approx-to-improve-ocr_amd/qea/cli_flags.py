"""The reference's command-line surface, as data.  Names, defaults, types, choices and the
`store_false` quirk of --random_std are those of patch_cli.py:11-155 and area_cli.py:11-124
(SURVEY.md §5.6); the MI355X build only ADDS flags (marked new)."""

SUBSETS = ["random", "uniformCER", "uniformCERglobal", "randomglobal", "rangeCER", "uniformEntropy", "topKCER"]
WEIGHTGEN = ["levenshtein", "self_attention", "decaying"]

# (flag, kwargs, which CLIs carry it: "p" = patch_cli, "a" = area_cli)
FLAGS = [
    ("--batch_size", dict(type=int, default=32, help="input batch size"), "a"),
    ("--lr_crnn", dict(type=float, default=0.0001, help="CRNN learning rate, not used by adadealta"), "pa"),
    ("--scalar", dict(type=float, default=1, help="scalar in which the secondary loss is multiplied"), "pa"),
    ("--lr_prep", dict(type=float, default=0.00005, help="prep model learning rate, not used by adadealta"), "pa"),
    ("--epoch", dict(type=int, default={"p": 25, "a": 50}, help="number of epochs"), "pa"),
    ("--random_seed", dict(type=int, default=42, help="Random seed for experiment"), "pa"),
    ("--warmup_epochs", dict(type=int, default=0, help="number of warmup epochs"), "pa"),
    ("--std", dict(type=int, default=5, help="standard deviation of Gaussian noice added to images (this value devided by 100)"), "pa"),
    ("--inner_limit", dict(type=int, default=2, help="number of inner loop iterations in Alogorithm 1"), "pa"),
    ("--inner_limit_skip", dict(action="store_true", help="In the first inner limit loop, do NOT add noise to the image"), "pa"),
    ("--crnn_model", dict(help="specify non-default CRNN model location. By default, a new CRNN model will be used"), "pa"),
    ("--prep_model", dict(help="specify non-default Prep model location. By default, a new Prep model will be used"), "pa"),
    ("--exp_base_path", dict(default=".", help="Base path for experiment. Defaults to current directory"), "pa"),
    ("--ocr", dict(default="Tesseract", help="performs training labels from given OCR [Tesseract,EasyOCR,gvision,stub]"), "pa"),
    ("--random_std", dict(action="store_false", help="randomly selected integers from 0 upto given std value (devided by 100) will be used"), "pa"),
    ("--minibatch_subset", dict(choices=SUBSETS, help="Specify method to pick subset from minibatch."), "pa"),
    ("--minibatch_subset_prop", dict(default=0.5, type=float, help="If --minibatch_subset is provided, specify percentage of samples per mini-batch."), "pa"),
    ("--start_epoch", dict(type=int, default=0, help="Starting epoch. If loading from a ckpt, pass the ckpt epoch here."), "pa"),
    ("--data_base_path", dict(default=".", help="Base path training, validation and test data"), "pa"),
    ("--exp_name", dict(default={"p": "test_patch", "a": "test_area"}, help="Specify name of experiment (JVP Jitter, Sample Dropping Etc.)"), "pa"),
    ("--exp_id", dict(help="Specify unique experiment ID"), "pa"),
    ("--train_subset_size", dict(type=int, help="Subset of training size to use"), "pa"),
    ("--val_subset_size", dict(type=int, help="Subset of val size to use"), "pa"),
    ("--weight_decay", dict(type=float, default=5e-4, help="Weight Decay for the optimizer"), "p"),
    ("--cers_ocr_path", dict(help="Cer information json"), "pa"),
    ("--image_prop", dict(type=float, help="Proportion of images per epoch"), "p"),
    ("--discount_factor", dict(type=float, default=1, help="Discount factor for CER values"), "p"),
    ("--update_CRNN", dict(action="store_true", help="Update CRNN along with the preprocessor"), "p"),
    ("--window_size", dict(type=int, default=1, help="Window size if tracking is enabled"), "pa"),
    ("--query_dim", dict(type=int, default=32, help="Dimension of the query/key projection (attention weight generator)"), "p"),
    ("--emb_dim", dict(type=int, default=256, help="Character embedding dimension (attention weight generator)"), "p"),
    ("--attn_activation", dict(default="sigmoid", help="Activation of the attention weight generator"), "p"),
    ("--weightgen_method", dict(choices=WEIGHTGEN, default="decaying", help="Method for generating loss weights for tracking"), "pa"),
    ("--decay_factor", dict(type=float, default=0.7, help="Decay factor for decaying loss weight generation"), "pa"),
    ("--optim_crnn_path", dict(help="CRNN optimizer state to resume from"), "p"),
    ("--optim_prep_path", dict(help="Preprocessor optimizer state to resume from"), "p"),
    ("--pruning_artifact", dict(help="Name of the pruning artifact (subset of the training set)"), "p"),
    ("--lr_scheduler", dict(choices=["cosine"], help="Learning-rate scheduler for the CRNN"), "a"),
    # ---- new (additive) ----
    ("--synthetic_size", dict(type=int, help="[new] train on N synthetic samples instead of reading --data_base_path"), "pa"),
    ("--select_before_clean", dict(action="store_true", help="[new] Phase A: pick the TopKCER / random subset FIRST and run the cleaner only on "
                                                             "the picked images (the pick depends on names and CERs alone; eval-mode BatchNorm "
                                                             "makes every image's output independent of the rest of the minibatch)"), "a"),
    ("--per_shard_topk", dict(action="store_true", help="[new] data-parallel runs only: pick the TopKCER subset per GPU shard instead of "
                                                        "over the whole minibatch (not the reference's selection; saves one 32 KB all-gather)"), "a"),
    ("--docs_per_step", dict(type=int, default=1, help="[new] documents per optimiser step (the reference hard-codes one, train_nn_patch.py:37). "
                                                      "The N documents go through the cleaner as ONE batch with per-document BatchNorm statistics "
                                                      "and through the CRNN of Phase B as one batch; forward values equal N sequential passes, the "
                                                      "gradients of the N documents accumulate into one Adam step"), "p"),
    ("--graph", dict(action="store_true", help="[new] replay Phase B (cleaner -> CRNN -> CTC + MSE -> backward -> Adam) and the CRNN side of Phase A as ONE hipGraph each per "
                                               "(batch size, width): at the reference's batch sizes (tens of strips) a step is bound by the host "
                                               "time of ~600 launches, which the replay removes; the first two steps of a shape run eagerly, "
                                               "single-process runs only"), "a"),
    ("--no_rebalance_topk", dict(action="store_true", help="[new] data-parallel runs only: process every global TopKCER winner on the rank "
                                                           "that owns it instead of dealing the winners out in equal slices (one all-reduce of "
                                                           "k x 16 KB images); always so with --inner_limit_skip (label histories stay with the owner)"), "a"),
]


def build_parser(which, description):
    import argparse
    ap = argparse.ArgumentParser(description=description)
    for flag, kw, where in FLAGS:
        if which not in where:
            continue
        kw = dict(kw)
        if isinstance(kw.get("default"), dict):
            kw["default"] = kw["default"][which]
        ap.add_argument(flag, **kw)
    return ap
