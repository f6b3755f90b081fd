import sys
sys.path.insert(0, ".")
import diee_amd
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
print("forward time (us) by batch: auto-per-layer | fused 4-board | fused 2-board")
for G in (1024, 768, 512, 448, 384, 320, 256, 192, 160, 128, 96, 64):
    r = []
    for v in (1 << 20, 100, 101):
        vv = 0 if v == 1 << 20 else v
        # variant 0 = auto per-layer geometry with fused disabled via a huge threshold is not reachable here,
        # so use explicit ids: 2 / 6 / 5 by batch like the dispatcher
        if v == 1 << 20:
            vv = 2 if G > 320 else (6 if G > 80 else 5)
        r.append(e.conv_bench(G, vv, 20)[2])
    print(f"G={G:5d}  {r[0]:8.1f} {r[1]:8.1f} {r[2]:8.1f}")
