"""The engine and torch.cuda in ONE process, engine initialised first: PyTorch wheels bundle their own HIP runtime, and two runtimes in a
process leave torch without devices (tools/hip_coexist_probe.py). facet_amd/_lib.py maps torch's copy before the engine when one is
installed, so Facet's torch models (and torch.distributed / RCCL) keep working next to the engine whatever the import order."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu


def test_engine_first_then_torch_cuda_in_a_fresh_process():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\\n"
        "import numpy as np\\n"
        "assert 'torch' not in sys.modules\\n"
        "from facet_amd import Engine\\n"
        "e = Engine(0, arena_bytes=1 << 30)\\n"
        "st = e.image_stats(np.full((1, 16, 16, 3), 7, np.uint8))[0]\\n"
        "import torch\\n"
        "assert torch.cuda.is_available(), 'torch lost the GPU after the engine initialised HIP'\\n"
        "x = torch.arange(6, device='cuda', dtype=torch.float32)\\n"
        "assert float(x.sum()) == 15.0 and st[0, 7] == 256.0\\n"
        "assert e.image_stats(np.full((1, 16, 16, 3), 9, np.uint8))[0][0, 9] == 256.0\\n"
        "print('coexist ok')\\n" % root)
    r = subprocess.run([sys.executable, "-c", code.replace("\\n", "\n")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "coexist ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
