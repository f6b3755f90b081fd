"""Pair tower with its k loop unrolled in full (-DDIEE_PAIR_UNROLL=1 -> libdiee_pairu.so) against the product, forward latency in us
(variant 110 = k_tower16p<4>, 111 = <2>), each build in a fresh process:
    DIEE_OUT=libdiee_pairu.so DIEE_EXTRA_FLAGS=-DDIEE_PAIR_UNROLL=1 python die-e_amd/build.py; python scripts/pair_unroll_ab.py"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, root)
    import diee_amd
    e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
    print("   ".join(f"G={G}: " + " ".join(f"{e.conv_bench(G, v, 60)[2]:6.1f}" for _ in range(2)) for G, v in ((300, 110), (400, 110), (512, 110), (200, 111), (256, 111))))
    sys.exit(0)
for rep in range(2):
    for lib, what in (("libdiee.so", "product"), ("libdiee_pairu.so", "k loop unrolled")):
        env = dict(os.environ); env["DIEE_LIB"] = os.path.join(root, "die-e_amd", lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, cwd=root, capture_output=True, text=True)
        print(f"{what:18s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)
