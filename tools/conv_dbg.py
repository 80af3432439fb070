import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facet_amd import Engine
eng = Engine(0, arena_bytes=16 << 30)
mb=32
for name,(h,cin,cout,k,s,p) in {"3x3 256->256@64":(64,256,256,3,1,1),"1x1 1024->256@64":(64,1024,256,1,1,0)}.items():
    fl = 2.0 * mb * (h // s) ** 2 * cin * k * k * cout
    for v in (11, 811, 11, 811):
        ms = eng.bench_conv(mb, h, h, cin, cout, k, s, p, False, "relu", v, 5)
        print(f"{name} variant={v}: {fl/ms/1e9:.1f} TF/s")
