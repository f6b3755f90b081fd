import sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G in (1024, 900, 768, 640, 512, 448, 384, 300):
    row = [f"G {G:5d}"]
    for v in (102, 105, 106, 108, 103):
        row.append(f"g{v - 100} {e.conv_bench(G, v, 30)[2]:7.1f}")
    print("  ".join(row), flush=True)
