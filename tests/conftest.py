import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cache = {}

    def load(name):
        if name not in cache:
            with open(os.path.join(gdir, name + ".json")) as fh:
                cache[name] = json.load(fh)
        return cache[name]

    return load
