"""Stand-in model files for tests, bench.py and tools - NOT part of the product package.

The reference's InsightFace stage loads three ONNX files (buffalo_l: det_10g.onnx, 2d106det.onnx, w600k_r50.onnx, fetched by
insightface at analyzers/face.py:30-38) that do not exist offline. `synthetic_onnx` builds seeded graphs of the same three
architectures and `onnx_writer` serialises them as genuine .onnx bytes, so the engine's ONNX runtime (facet_amd/csrc/onnx_graph.hip)
is exercised through exactly the interface it serves real model files with. Nothing under facet_amd/ imports this package.
"""
