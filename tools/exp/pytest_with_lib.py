#!/usr/bin/env python3
"""pytest against another build of libhevcdbk.so: pytest_with_lib.py path/to/lib.so [pytest arguments]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import gpu_video_codec_amd._lib as l
l.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
raise SystemExit(pytest.main(sys.argv[2:]))
