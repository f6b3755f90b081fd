import sys
sys.path.insert(0, ".")
import diee_amd
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
vs = [int(x) for x in sys.argv[1].split(",")]
Gs = [int(x) for x in sys.argv[2].split(",")]
print("forward us for fused geometries", vs)
for G in Gs:
    r = [e.conv_bench(G, v, 20)[2] for v in vs]
    print(f"G={G:5d}  " + " ".join(f"{x:8.1f}" for x in r))
