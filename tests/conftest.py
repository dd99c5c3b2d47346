import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Accessor for tests/golden/rust_arrays.json (made by tests/golden/extract_vectors.py)."""

    def __init__(self):
        with open(os.path.join(ROOT, "tests", "golden", "rust_arrays.json")) as fh:
            self.d = json.load(fh)

    def get(self, file_suffix, name, idx=0, fn=None):
        files = [f for f in self.d if f.endswith(file_suffix)]
        assert len(files) == 1, files
        hits = []
        for k, v in self.d[files[0]].items():
            kfn, kname, kidx = k.rsplit(":", 2)
            if kname == name and int(kidx) == idx and (fn is None or kfn == fn):
                hits.append(v)
        assert len(hits) == 1, (file_suffix, name, idx, len(hits))
        return hits[0]

    def bytes(self, file_suffix, name, idx=0, fn=None):
        return bytes(self.get(file_suffix, name, idx, fn))


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o
