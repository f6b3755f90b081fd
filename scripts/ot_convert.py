#!/usr/bin/env python3
"""Converter between die-e's libtorch archives and this engine's files:

    python scripts/ot_convert.py model best_model.ot best_model.npy     # .ot -> blob (.npy) or the reverse, by extension
    python scripts/ot_convert.py data  sp-0/ sp-0-npy/ --to npy         # ps/states/outcomes .ot -> .npy (or --to ot)
"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ot = importlib.import_module("die-e_amd.ot")

ap = argparse.ArgumentParser()
sub = ap.add_subparsers(dest="what", required=True)
m = sub.add_parser("model"); m.add_argument("src"); m.add_argument("dst")
d = sub.add_parser("data"); d.add_argument("src"); d.add_argument("dst"); d.add_argument("--to", choices=["ot", "npy"], required=True)
a = ap.parse_args()
if a.what == "model":
    blob = ot.load_model(a.src)
    if a.dst.endswith(".ot"):
        ot.save_model_ot(blob, a.dst)
    else:
        np.save(a.dst, blob)
    print(f"{a.src} -> {a.dst}: {blob.size} parameters")
else:
    ot.convert_data_dir(a.src, a.dst, a.to)
    print(f"{a.src} -> {a.dst} ({a.to})")
