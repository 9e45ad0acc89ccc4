#!/usr/bin/env python3
"""Run bench.py against another build of libhevcdbk.so (same-box A/B of two kernel versions):
   python tools/bench_with_lib.py path/to/libhevcdbk.so [bench.py arguments]"""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import gpu_video_codec_amd._lib as l
l.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
