import sys
sys.path.insert(0, ".")
import diee_amd
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
print("forward us: fused32 4b | fused32 2b | fused16 4b | fused16 2b | fused16 3b")
for G in (1024, 896, 768, 640, 512, 384, 320, 256, 192, 128):
    r = [e.conv_bench(G, v, 20)[2] for v in (100, 101, 102, 103, 104)]
    print(f"G={G:5d}  " + " ".join(f"{x:8.1f}" for x in r))
