"""Development: device time of the tower conv kernel per geometry and batch size."""
import sys
sys.path.insert(0, ".")
import diee_amd
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
print("G variant us_mode0 us_mode1 us_forward TF(mode0)")
for G in (1024, 768, 512, 384, 256, 192, 128, 96, 64, 32, 8):
    for v in (1, 2, 3, 5, 6, 7):
        if v == 1 and G < 64: continue
        a, b, f = e.conv_bench(G, v, 30)
        print(f"{G:5d} {v} {a:8.1f} {b:8.1f} {f:9.1f}  {2*G*24*2304*256/a/1e6:7.1f}")
