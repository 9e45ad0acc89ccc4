import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def sha256(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_inputs(manifest):
    out = {}
    for name, ent in manifest["images"].items():
        with open(os.path.join(GOLDEN, ent["file"]), "rb") as fh:
            out[name] = fh.read()
        assert sha256(out[name]) == ent["input_sha256"]
    return out


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o
