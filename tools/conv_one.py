"""Run ONE conv shape a few times (for rocprofv3 --pmc runs). args: h cin cout k stride pad res variant mb"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
h, cin, cout, k, s, p, res, v, mb = [int(a) for a in sys.argv[1:10]]
eng = Engine(0, arena_bytes=16 << 30)
ms = eng.bench_conv(mb, h, h, cin, cout, k, s, p, bool(res), "relu", v, 3)
fl = 2.0 * mb * (h // s) ** 2 * cin * k * k * cout
print(f"{ms:.4f} ms  {fl/ms/1e9:.1f} TF/s")
