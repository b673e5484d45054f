"""Loader for the committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.npz = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(bytes(self.npz["meta"]).decode())

    def group(self, prefix):
        """dict of torch tensors for keys '<prefix>/<name>' (names may contain dots, never slashes)."""
        pre = prefix + "/"
        out = {}
        for k in self.npz.files:
            if k.startswith(pre) and "/" not in k[len(pre):]:
                out[k[len(pre):]] = torch.from_numpy(np.asarray(self.npz[k]))
        return out

    def get(self, key):
        return torch.from_numpy(np.asarray(self.npz[key]))

    def has(self, key):
        return key in self.npz.files


def rel_err(a, b):
    """norm-wise relative error ||a-b|| / max(||b||, tiny) in float64."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))
