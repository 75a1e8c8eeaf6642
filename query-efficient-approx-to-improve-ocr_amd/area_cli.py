"""`python area_cli.py ...` — front-end of the text-area trainer with the reference's flags
(area_cli.py:9-141): parses, writes params.txt into the experiment directory, trains."""
import datetime
import os

import properties
from qea.cli_flags import build_parser

if __name__ == "__main__":
    args = build_parser("a", "Trains the Prep with text-area (VGG / POS strip) datasets").parse_args()
    print(vars(args))
    from train_nn_area import TrainNNPrep
    start = datetime.datetime.now()
    trainer = TrainNNPrep(args)
    with open(os.path.join(args.exp_base_path, properties.param_path), "w") as f:
        f.write(f"{vars(args)}\nStart:{start}\n")
    trainer.train()
    with open(os.path.join(args.exp_base_path, properties.param_path), "a") as f:
        f.write(f"End:{datetime.datetime.now()}\n")
