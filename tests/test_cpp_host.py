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
    # once as it ships, once with EncodeBatchFlat's sub-batches cut at 30 KB (128 MB in production: the gather of one sub-batch overlaps the
    # device work of the one before it)
    for env in (None, dict(os.environ, TKZ_FLAT_SUBBATCH_BYTES="30000")):
        out = subprocess.run([exe, str(vocab), os.path.join(GOLDEN, "lib.rs.txt"), os.path.join(GOLDEN, "tokens_gpt2.json")],
                             capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "cpp host mirror ok" in out.stdout


def ids_checksum(ids):
    """The position-weighted sum mod 2^64 that tests/cpp/bench_host_api.cpp prints."""
    import numpy as np
    a = np.asarray(ids).astype(np.uint32).astype(np.uint64) + np.uint64(1)
    w = np.arange(len(a), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
    return "%016x" % int((a * w).sum(dtype=np.uint64))


def build_host_api_bench(outdir, libdir, libname):
    exe = os.path.join(str(outdir), "bench_host_api")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "bench_host_api.cpp"),
                           "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_host_api_bench_driver_on_emulated_kernels(tmp_path):
    """bench.py's `value_host_api` leg (tests/cpp/bench_host_api.cpp: TikTokenizer::EncodeBatchFlat on std::strings read from a sample
    file) gives the ids the Python binding gives, here on the CPU-emulated kernels."""
    import json
    import numpy as np
    import emu
    import parity
    from tokenizer_amd import _native as N
    from tokenizer_amd import tokenizer as TK
    lib = emu.library()
    raw = gzip.decompress(open(os.path.join(GOLDEN, "gpt2.tiktoken.gz"), "rb").read())
    (tmp_path / "v.tiktoken").write_bytes(raw)
    (tmp_path / "regex.txt").write_text(TK.REGEX_CL100K)
    docs = [N.corpus_doc_host(1, 0x5EED0002, d, 64, 400, lib=lib) for d in range(150)]
    data, offs = parity.pack(docs)
    with open(tmp_path / "sample.bin", "wb") as f:
        f.write(np.int64(len(docs)).tobytes()); f.write(offs.astype(np.int64).tobytes()); f.write(data.tobytes())
    ids, _ = N.Encoder(N.Vocab(raw, lib), N.CL100K).encode_batch(data, offs)
    exe = build_host_api_bench(tmp_path, os.path.dirname(emu.EMU_LIB), "tkz_hostemu")
    for env in (None, dict(os.environ, TKZ_FLAT_SUBBATCH_BYTES="9000")):          # (in one call; in three pipelined sub-batches)
        out = subprocess.run([exe, str(tmp_path / "v.tiktoken"), str(tmp_path / "regex.txt"), str(tmp_path / "sample.bin"), "2"], capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        r = json.loads(out.stdout)
        assert r["docs"] == 150 and r["bytes"] == len(data) and r["tokens"] == len(ids) and r["ids_checksum"] == ids_checksum(ids)


def test_cpp_host_mirror_on_emulated_kernels(tmp_path):
    import emu
    emu.library()
    _build_and_run(tmp_path, os.path.dirname(emu.EMU_LIB), "tkz_hostemu")


@pytest.mark.gpu
def test_cpp_host_mirror_on_gpu(tmp_path):
    _build_and_run(tmp_path, os.path.join(ROOT, "tokenizer_amd", "lib"), "tkz")
