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
    These restate the published architectures -> "parity unpinned" against those packages. Second opinion (not the
    reference's code path, but independent code): the ResNet-50 pyramid and both CLIP towers equal HuggingFace
    transformers' implementations with the same weights (tests/test_oracle_second_opinion.py); the CFANet head has none.
  * ONNX graph evaluation (oracle/onnx_ref.py), face pre/post-processing (oracle/face_ref.py), technical metrics
    (oracle/technical_ref.py), leading lines (oracle/lines_ref.py: Gaussian / Canny / probabilistic Hough): restate ONNX operator semantics / insightface / OpenCV fixed-point arithmetic
    [DEP-KNOWLEDGE]; onnxruntime, insightface, cv2 and the buffalo_l files are absent -> "parity unpinned"; the OpenCV
    pieces are held to known answers in tests/test_cv_semantics.py and tests/test_lines_host.py; the ONNX evaluator reproduces
    torch's outputs on models serialised by PyTorch's own exporter (tests/golden/torch_onnx_*.npz).
"""
