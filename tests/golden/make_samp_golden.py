"""Generates tests/golden/samp_golden.npz from the REFERENCE's own classes (run in the build container only).

Imports /root/reference/models/samp_net.py with a stub `torchvision` (the module needs it only for
transforms.* and models.resnet18, samp_net.py:18-19,659,823-830) and instantiates the nn.Module classes
directly — never SAMPNetScorer / SaliencyDetector, whose constructors download weights (samp_net.py:394,854).
Weights are the seeded synthetic checkpoints of facet_amd.weights (regenerated from the seed, not committed);
inputs are seeded. Outputs (the reference's U2NETP d0 and SAMPNet (pattern_weights, attributes, score_dist))
are stored as the golden vectors that pin oracle/sampnet.py and the HIP path.

Usage:  python tests/golden/make_samp_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from facet_amd.weights import synthetic_state_dict  # noqa: E402
from oracle.resnet import _FakeTorchvisionResNet18  # noqa: E402

SEED_W = 7
SEED_X = 11


def import_reference():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet18 = lambda weights=None: _FakeTorchvisionResNet18()
    tvm.ResNet18_Weights = types.SimpleNamespace(DEFAULT=None)
    tv.transforms, tv.models = tvt, tvm
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.models": tvm})
    sys.path.insert(0, REF)
    from models import samp_net  # the reference module
    return samp_net


def golden_inputs(n=2):
    g = torch.Generator().manual_seed(SEED_X)
    return torch.randn(n, 3, 224, 224, generator=g)


def main():
    ref = import_reference()
    torch.manual_seed(0)
    u2 = ref.U2NETP(3, 1).eval()
    sn = ref.SAMPNet().eval()
    sd_u = {k: torch.from_numpy(v) for k, v in synthetic_state_dict("u2netp", SEED_W).items()}
    sd_s = {k: torch.from_numpy(v) for k, v in synthetic_state_dict("samp_net", SEED_W).items()}
    # BatchNorm's num_batches_tracked is the only buffer the synthetic checkpoint does not carry.
    miss_u = u2.load_state_dict(sd_u, strict=False)
    miss_s = sn.load_state_dict(sd_s, strict=False)
    for miss in (miss_u, miss_s):
        assert not miss.unexpected_keys, miss.unexpected_keys
        assert all(k.endswith("num_batches_tracked") for k in miss.missing_keys), miss.missing_keys
    x = golden_inputs()
    with torch.no_grad():
        sal = u2(x)[0]
        pw, attrs, dist = sn(x, sal)
        fm = sn.backbone(x)
    out = os.path.join(ROOT, "tests", "golden", "samp_golden.npz")
    np.savez_compressed(
        out, seed_w=SEED_W, seed_x=SEED_X,
        saliency_ds=sal[:, 0, ::8, ::8].numpy(),            # 28x28 subsample of d0
        saliency_mean=sal.mean(dim=(1, 2, 3)).numpy(), saliency_sum_abs=sal.abs().sum().item(),
        saliency_row100=sal[:, 0, 100, :].numpy(),
        feature_map_mean=fm.mean(dim=(2, 3)).numpy(),      # [2,512]
        pattern_weights=pw.numpy(), attributes=attrs.numpy(), score_dist=dist.numpy())
    print("wrote", out, {k: v.shape for k, v in dict(pattern_weights=pw, attributes=attrs, score_dist=dist).items()})
    print("pw", pw.numpy()[0], "\ndist", dist.numpy())


if __name__ == "__main__":
    main()
