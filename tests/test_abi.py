"""`not gpu`: the C-ABI library builds for gfx950, loads, and exports every symbol include/tkz.h
declares; host-only entry points (vocabulary loading) behave like the reference's loader; and without
a GPU the encoder entry points fail loudly (there is no CPU fallback)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT
from tokenizer_amd import _native as N


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tokenizer_amd", "csrc"), "-s"])
    return N.Library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tkz.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tkz_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib.path], capture_output=True, text=True, check=True).stdout
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    decl = declared_symbols()
    assert len(decl) >= 20
    missing = [s for s in decl if s not in exported]
    assert not missing, missing
    for s in decl:
        getattr(lib.L, s)


def test_device_code_object_is_gfx950(lib):
    out = subprocess.run(["strings", "-n", "6", lib.path], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_vocab_loader_host_side(lib, gpt2_tiktoken_bytes):
    v = N.Vocab(gpt2_tiktoken_bytes, lib)
    assert len(v) == 50256 and v.max_key_len == 128 and v.pair_table_entries == 108299
    assert v.rank(b"!") == 0 and v.rank(b" the") == 262 and v.rank(b"\xff\xfe") == -1
    tb = v.table_bytes()                           # the whole table set is sized for one XCD's 4 MiB L2
    assert tb["total"] == tb["short"] + tb["mid"] + tb["long"] + tb["pair"] + tb["direct"] and 1 << 20 < tb["total"] < 4 << 20
    with pytest.raises(N.DuplicateRankError):      # ArgumentException, TikTokenizer.cs:84-87
        N.Vocab(b"YQ== 0\nYg== 0\n", lib)
    with pytest.raises(N.FormatError):             # TikTokenizer.cs:114-118
        N.Vocab(b"YQ== 0 1\n", lib)
    with pytest.raises(N.FormatError):             # TikTokenizer.cs:122-129
        N.Vocab(b"YQ== x\n", lib)
    with pytest.raises(N.FormatError):
        N.Vocab(b"Y!== 1\n", lib)
    v2 = N.Vocab(b"\nYQ== 0\n\r\n  \nYg== 1\n", lib)     # blank lines skipped, TikTokenizer.cs:109-112
    assert len(v2) == 2 and v2.rank(b"b") == 1
    # the reference reads through a UTF-8 StreamReader: a line of U+0085 / U+00A0 / U+3000 is blank, a lone byte 0x85 is U+FFFD
    assert len(N.Vocab(b"YQ== 0\n\xc2\x85\xc2\xa0 \xe3\x80\x80\nYg== 1\n", lib)) == 2
    with pytest.raises(N.FormatError):
        N.Vocab(b"YQ== 0\n\x85\nYg== 1\n", lib)


def test_pattern_from_regex(lib):
    import ctypes as C
    out = C.c_int32(0)
    p1 = rb"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"
    cl = rb"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
    assert lib.L.tkz_pattern_from_regex(p1, C.byref(out)) == 0 and out.value == N.P1
    assert lib.L.tkz_pattern_from_regex(cl, C.byref(out)) == 0 and out.value == N.CL100K
    assert lib.L.tkz_pattern_from_regex(rb"\w+", C.byref(out)) == N.E_UNSUPPORTED


def test_no_gpu_means_loud_failure(lib, gpt2_tiktoken_bytes):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    v = N.Vocab(gpt2_tiktoken_bytes, lib)
    with pytest.raises(N.TkzError) as ei:
        N.Encoder(v, N.CL100K)
    assert ei.value.code == N.E_NO_DEVICE


def test_product_package_does_not_reference_oracle_or_emulator():
    pkg = os.path.join(ROOT, "tokenizer_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                s = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in s.replace("no oracle", "") or f in ("tkz_simt.h",) or "import oracle" not in s
                assert "import oracle" not in s and "from oracle" not in s and "libtkz_hostemu" not in s
