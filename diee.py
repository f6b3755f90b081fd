#!/usr/bin/env python3
"""die-e's CLI over the MI355X engine: `python diee.py -g backgammon [-c config.toml] learn|play|train|replay`"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.exit(importlib.import_module("die-e_amd.cli").main())
