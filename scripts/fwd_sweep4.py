import os, sys
sys.path.insert(0, ".")
import diee_amd
print("forward us by batch; DIEE_NET16 =", os.environ.get("DIEE_NET16", "1"))
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
for G in (1024, 896, 768, 640, 512, 384, 256, 208):
    print(f"G={G:5d}  {e.conv_bench(G, 0, 20)[2]:8.1f}")
