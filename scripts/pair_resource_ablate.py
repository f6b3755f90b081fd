"""What bounds the pair tower: timing builds (wrong results) of k_tower16p without its hand-off / LDS reads / weight loads.
    DIEE_OUT=libdiee_pa_x2.so   DIEE_EXTRA_FLAGS=-DDIEE_PAIR_ABLATE=2                   no exchange at all
    DIEE_OUT=libdiee_pa_r1.so   DIEE_EXTRA_FLAGS=-DDIEE_PAIR_RES=1                      A fragments read on every second k-step only
    DIEE_OUT=libdiee_pa_r2.so   ... =2 never, _r3 =3 no weight loads, _r4 =4 neither,  _r4x2: -DDIEE_PAIR_RES=4 -DDIEE_PAIR_ABLATE=2 (MFMAs + epilogue only)
Each build in a fresh process: forward latency (us) of variant 110 (<4>) at 300 / 400 / 512 boards and 111 (<2>) at 200 boards."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, root)
    import diee_amd
    e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
    out = []
    for G, v in ((300, 110), (400, 110), (512, 110), (200, 111)):
        out.append(f"G={G}: " + " ".join(f"{e.conv_bench(G, v, 60)[2]:6.1f}" for _ in range(3)))
    print("   ".join(out)); sys.exit(0)
names = {"libdiee.so": "product build", "libdiee_pa_x2.so": "no exchange", "libdiee_pa_r1.so": "A fragments on every second k-step",
         "libdiee_pa_r2.so": "no A-fragment reads", "libdiee_pa_r3.so": "no weight loads", "libdiee_pa_r4.so": "neither",
         "libdiee_pa_r4x2.so": "neither, no exchange (MFMAs + epilogue)"}
for lib, what in names.items():
    path = os.path.join(root, "die-e_amd", lib)
    if not os.path.exists(path): continue
    env = dict(os.environ); env["DIEE_LIB"] = path
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, cwd=root, capture_output=True, text=True)
    print(f"{what:42s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)
