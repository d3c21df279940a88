import os

import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    fx = torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), weights_only=False)
    fx["state"] = torch.load(os.path.join(GOLDEN_DIR, fx["state_file"]), weights_only=False)
    return fx
