"""A/B timing of conv tile variants on the ResNet-50@1024^2 layer shapes (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 4]
eng = Engine(0, arena_bytes=24 << 30)
SHAPES = [  # name, h, cin, cout, k, stride, pad, res
    ("l1.conv2 3x3 64->64 @256", 256, 64, 64, 3, 1, 1, False),
    ("l1.conv3 1x1 64->256 +res", 256, 64, 256, 1, 1, 0, True),
    ("l1.conv1 1x1 256->64", 256, 256, 64, 1, 1, 0, False),
    ("l2.conv2 3x3 128->128 @128", 128, 128, 128, 3, 1, 1, False),
    ("l2.conv3 1x1 128->512 +res", 128, 128, 512, 1, 1, 0, True),
    ("l3.conv2 3x3 256->256 @64", 64, 256, 256, 3, 1, 1, False),
    ("l3.conv3 1x1 256->1024 +res", 64, 256, 1024, 1, 1, 0, True),
    ("l3.conv1 1x1 1024->256", 64, 1024, 256, 1, 1, 0, False),
    ("l4.conv2 3x3 512->512 @32", 32, 512, 512, 3, 1, 1, False),
    ("l4.conv3 1x1 512->2048 +res", 32, 512, 2048, 1, 1, 0, True),
    ("head 1x1 2048->2048 @32", 32, 2048, 2048, 1, 1, 0, False),
    ("head 1x1 64->64 @512", 512, 64, 64, 1, 1, 0, False),
]
print(f"mb={mb}; TF/s per variant {variants}")
for name, h, cin, cout, k, s, p, res in SHAPES:
    fl = 2.0 * mb * (h // s) ** 2 * cin * k * k * cout
    row = []
    for v in variants:
        try:
            ms = eng.bench_conv(mb, h, h, cin, cout, k, s, p, res, "relu", v, 5)
            row.append(f"{fl / ms / 1e9:7.1f}")
        except Exception as e:
            row.append("   n/a "); print(e)
    print(f"{name:32s} " + " ".join(row))
