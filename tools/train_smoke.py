"""Developer tool: a few epochs of the area trainer on synthetic data (HIP path) — loss must go down."""
import json, os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd")]
from datasets.synthetic import SyntheticTextAreas
from qea.cli_flags import build_parser
from train_nn_area import TrainNNPrep
tmp = tempfile.mkdtemp()
tr = SyntheticTextAreas(512, seed=1, include_name=True, include_index=True)
cers = os.path.join(tmp, "cers.json"); json.dump({n: 1.0 for n in tr.names}, open(cers, "w"))
args = build_parser("a", "").parse_args(["--exp_base_path", os.path.join(tmp, "exp"), "--ocr", "stub", "--epoch", "6", "--batch_size", "64",
                                          "--minibatch_subset", "topKCER", "--minibatch_subset_prop", "0.9", "--cers_ocr_path", cers,
                                          "--inner_limit", "2", "--lr_crnn", "0.001", "--lr_prep", "0.0005"])
t = TrainNNPrep(args, train_set=tr, val_set=SyntheticTextAreas(64, seed=2, include_name=True))
t.train()
