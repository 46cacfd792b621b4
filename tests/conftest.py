"""Shared pytest configuration: registers the `gpu` marker and common paths."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden", "integration_test_suite")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def read_expected_tree(root):
    """{relative_path: sorted list of 4-line FASTQ record texts} of a specimux output tree."""
    tree = {}
    for dirpath, _dirs, files in os.walk(root):
        for fn in files:
            if not fn.endswith(".fastq"):
                continue
            full = os.path.join(dirpath, fn)
            with open(full) as fh:
                lines = fh.read().split("\n")
            recs = ["\n".join(lines[i:i + 4]) + "\n" for i in range(0, len(lines) - 1, 4)]
            tree[os.path.relpath(full, root)] = sorted(recs)
    return tree
