"""The C++ host mirror (include/tkz_tokenizer.hpp) compiled with g++ and run through the C ABI: on CPU against the
emulated build of the kernels, on the GPU box against libtkz.so."""
import gzip
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT


def _build_and_run(tmp_path, libdir, libname):
    vocab = tmp_path / "gpt2.tiktoken"
    vocab.write_bytes(gzip.decompress(open(os.path.join(GOLDEN, "gpt2.tiktoken.gz"), "rb").read()))
    exe = str(tmp_path / "test_tokenizer")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_tokenizer.cpp"),
                           "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe, str(vocab), os.path.join(GOLDEN, "lib.rs.txt"), os.path.join(GOLDEN, "tokens_gpt2.json")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpp host mirror ok" in out.stdout


def test_cpp_host_mirror_on_emulated_kernels(tmp_path):
    import emu
    emu.library()
    _build_and_run(tmp_path, os.path.dirname(emu.EMU_LIB), "tkz_hostemu")


@pytest.mark.gpu
def test_cpp_host_mirror_on_gpu(tmp_path):
    _build_and_run(tmp_path, os.path.join(ROOT, "tokenizer_amd", "lib"), "tkz")
