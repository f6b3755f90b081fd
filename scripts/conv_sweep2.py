import sys
sys.path.insert(0, ".")
import diee_amd
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
vs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2]
Gs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 768, 512, 384]
for G in Gs:
    for v in vs:
        a, b, f = e.conv_bench(G, v, 40)
        print(f"G={G:5d} v={v} mode0 {a:6.1f} us  mode1 {b:6.1f} us  forward {f:7.1f} us  {2*G*24*2304*256/a/1e6:7.1f} TF")
