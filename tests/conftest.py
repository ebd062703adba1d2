import gzip
import json
import os
import sys

import pytest

try:                      # torch (bundled HIP runtime) must initialise before libtkz is loaded, see tokenizer_amd/_native.py
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpt2_tiktoken_bytes():
    with open(os.path.join(GOLDEN, "gpt2.tiktoken.gz"), "rb") as f:
        return gzip.decompress(f.read())


# Rank files in tests/golden/: gpt2 is the one vocabulary the reference ships (model/gpt2.tiktoken); synth100k / synth200k are
# stand-ins of the SIZE of cl100k_base / o200k_base (which the reference downloads at run time and which exist nowhere offline),
# trained by tools/train_bpe.py with the cl100k / o200k split pattern: 100,256 and 199,998 keys, keys up to 128 bytes, ids > 2^17.
VOCAB_FILES = {"gpt2": "gpt2.tiktoken.gz", "synth100k": "synth100k.tiktoken.gz", "synth200k": "synth200k.tiktoken.gz"}


@pytest.fixture(scope="session")
def vocab_bytes():
    cache = {}

    def get(name):
        if name not in cache:
            with open(os.path.join(GOLDEN, VOCAB_FILES[name]), "rb") as f:
                cache[name] = gzip.decompress(f.read())
        return cache[name]
    return get


@pytest.fixture(scope="session")
def lib_rs_bytes():
    with open(os.path.join(GOLDEN, "lib.rs.txt"), "rb") as f:
        return f.read()


def load_golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def find_vocab_file(name):
    """cl100k_base / p50k_base / o200k_base are downloaded by the reference at run time
    (TokenizerBuilder.cs:113,129,141,156,168) and are not in this repo; tests that need them look in
    $TKZ_VOCAB_DIR and skip when absent."""
    d = os.environ.get("TKZ_VOCAB_DIR")
    if d:
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    return None


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def oracle_gpt2(oracle_mod, gpt2_tiktoken_bytes):
    return oracle_mod.Vocab(gpt2_tiktoken_bytes)
