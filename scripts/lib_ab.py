"""Forward latency (us, diee_dev_conv_bench) of several builds of libdiee, each in a fresh process, interleaved and repeated:
    python scripts/lib_ab.py libdiee.so libdiee_epi.so -- 1024:105 768:114 520:106
(G:variant pairs: geometry ids of launch_tower + 100; 110 / 111 = the pair tower; 0 = the product's dispatch)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--one":
    sys.path.insert(0, root)
    import diee_amd
    e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
    out = []
    for gv in sys.argv[2:]:
        G, v = (int(x) for x in gv.split(":"))
        out.append(f"G={G}/{v}: " + " ".join(f"{e.conv_bench(G, v, 60)[2]:6.1f}" for _ in range(3)))
    print("   ".join(out)); sys.exit(0)
i = sys.argv.index("--")
libs, cases = sys.argv[1:i], sys.argv[i + 1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ); env["DIEE_LIB"] = os.path.join(root, "die-e_amd", lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"] + cases, env=env, cwd=root, capture_output=True, text=True)
        print(f"{lib:22s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)
