"""CPU oracle of the image-scoring hot path — TEST INFRASTRUCTURE ONLY.

Plain PyTorch-CPU fp32 restatements of the models the reference (rlorenzo/facet) runs on its hot path,
each citing the reference file:line (or the un-vendored third-party architecture) it follows. Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as
the checker / the timed CPU baseline — never the product path (facet_amd/ must not import it).

Pinning status (see DESIGN.md "Oracle"):
  * U2NETP + SAMPNet (oracle/sampnet.py): PINNED — checked against the reference's own classes
    (models/samp_net.py imported in the build container with a stub torchvision) and against the golden
    vectors in tests/golden/ that import produced (tests/golden/make_samp_golden.py).
  * ResNet-50 backbone + CFANet head (oracle/topiq.py), CLIP ViT-L/14 (oracle/clip_vit.py): the arithmetic
    lives in pyiqa / timm / open_clip, which are NOT vendored in the reference and not installed here
    (requirements.txt:8,36, lower-bound pins only); the reference holds no tests or fixtures for them.
    These restate the published architectures -> "parity unpinned".
"""
