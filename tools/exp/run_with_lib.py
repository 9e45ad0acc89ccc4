#!/usr/bin/env python3
"""Run a tool against another build of libhevcdbk.so: run_with_lib.py path/to/lib.so tools/x.py [arguments]"""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import gpu_video_codec_amd._lib as l
l.LIB_PATH = os.path.abspath(sys.argv[1])
script = os.path.abspath(sys.argv[2])
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
