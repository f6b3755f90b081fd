"""What bounds the 4-board fused tower: timing builds (wrong results) without the loop's LDS reads / weight loads.
    for n in 6 7 8 9; do DIEE_OUT=libdiee_abl$n.so DIEE_EXTRA_FLAGS=-DDIEE_TOWER_ABLATE=$n python die-e_amd/build.py; done
    DIEE_OUT=libdiee_dupw.so DIEE_EXTRA_FLAGS=-DDIEE_TOWER_DUPW=1 python die-e_amd/build.py
    DIEE_OUT=libdiee_dupw7.so DIEE_EXTRA_FLAGS="-DDIEE_TOWER_DUPW=1 -DDIEE_TOWER_ABLATE=7" python die-e_amd/build.py
    python scripts/fused_resource_ablate.py
Each build in a fresh process: forward latency (us) of <4,8,3> at 1024 boards and <4,8,6> at 768 / 520 boards."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, root)
    import diee_amd
    e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
    out = []
    for G, v in ((1024, 108), (768, 106), (520, 106), (2048, 108)):
        out.append(f"G={G}: " + " ".join(f"{e.conv_bench(G, v, 60)[2]:6.1f}" for _ in range(3)))
    print("   ".join(out)); sys.exit(0)
names = {"libdiee.so": "product build", "libdiee_abl7.so": "7: A fragments read on every second k-step only", "libdiee_abl6.so": "6: no A-fragment reads in the loop",
         "libdiee_abl8.so": "8: no weight loads in the loop", "libdiee_abl9.so": "9: neither (MFMAs + epilogue + barrier only)",
         # round 4: the traffic of a 2 row-group x 4 column-group split of the workgroup's waves, priced without building it: every wave
         # also requests (and waits for) the weight fragments of wave ^ 4 (DIEE_TOWER_DUPW=1), with and without half the A-fragment LDS reads
         "libdiee_abl12.so": "12: weight fragments requested on every second k-step only (8 boards per workgroup's weight traffic)",
         "libdiee_dupw.so": "DUPW: weight fragments requested twice per CU", "libdiee_dupw7.so": "DUPW + 7: 2 x 4 split's traffic (half the A reads, twice the weight requests)"}
for lib, what in names.items():
    path = os.path.join(root, "die-e_amd", lib)
    if not os.path.exists(path): continue
    env = dict(os.environ); env["DIEE_LIB"] = path
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, cwd=root, capture_output=True, text=True)
    print(f"{what:55s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)
