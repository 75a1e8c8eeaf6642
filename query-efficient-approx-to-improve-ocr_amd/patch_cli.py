"""`python patch_cli.py ...` — front-end of the POS-patch trainer with the reference's flags
(patch_cli.py:9-176): parses, writes params.txt into the experiment directory, trains."""
import datetime
import os

import properties
from qea.cli_flags import build_parser

if __name__ == "__main__":
    args = build_parser("p", "Trains the Prep with Patch dataset").parse_args()
    print(vars(args))
    from train_nn_patch import TrainNNPrep
    start = datetime.datetime.now()
    trainer = TrainNNPrep(args)
    with open(os.path.join(args.exp_base_path, properties.param_path), "w") as f:
        f.write(f"{vars(args)}\nStart:{start}\n")
    trainer.train()
    with open(os.path.join(args.exp_base_path, properties.param_path), "a") as f:
        f.write(f"End:{datetime.datetime.now()}\n")
