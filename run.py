#!/usr/bin/env python3
"""Entry point with the reference's CLI (run.py:9-16):  python run.py -m SASRec -d amazon-toys
(single GPU), or under torch.distributed.run for single-node data parallelism over RCCL."""
import os

import torch

from dr4sr_amd import quickstart
from dr4sr_amd.utils import get_default_parser, load_config, setup_environment

if __name__ == "__main__":
    config = vars(get_default_parser().parse_args())
    config = load_config(config)
    setup_environment(config["train"])
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from dr4sr_amd.parallel import init_distributed
        torch.cuda.set_device(config["train"]["device"])
        init_distributed(config["train"]["device"])
    quickstart.run(config)
