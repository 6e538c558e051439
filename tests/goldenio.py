"""Loader for the committed golden fixtures (tests/golden/*.json.gz, see tools/gen_golden.py)."""
import gzip
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    with gzip.open(os.path.join(GOLD, name), 'rt') as f:
        return json.load(f)


def hx(s):
    return bytes.fromhex(s)
