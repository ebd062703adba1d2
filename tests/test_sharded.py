"""`not gpu`: the N > 1 path (contiguous document shards + ONE all-gather of the per-rank counts) with
world_size 2 over gloo.  Each rank encodes its shard with the emulated build of the kernels; rank 0
checks that shards concatenated == the whole batch (shard invariance) and that the gathered bases are right."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, q, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import gzip
    import torch.distributed as dist
    import emu
    from tokenizer_amd import _native as N
    from tokenizer_amd import sharded, write_shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = emu.library()
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "gpt2.tiktoken.gz"), "rb").read())
    enc = N.Encoder(N.Vocab(raw, lib), N.CL100K)
    n_total, lo_len, hi_len, seed = 301, 20, 300, 0x5EED0002
    lo, hi = sharded.shard_range(n_total, rank, world, lib=lib)
    docs = [N.corpus_doc_host(1, seed, d, lo_len, hi_len, lib=lib) for d in range(lo, hi)]
    data = np.frombuffer(b"".join(docs), np.uint8)
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.int64)
    ids, ooffs = enc.encode_batch(data, offs)
    g = sharded.gather_counts(hi - lo, int(offs[-1]), len(ids), lib=lib)
    write_shard(os.path.join(outdir, "tokens.%05d.tkzs" % rank), ids, ooffs, g["doc_base"], g["token_base"])
    q.put((rank, ids.tolist(), ooffs.tolist(), g["doc_base"], g["token_base"], g["docs"], g["bytes"], g["tokens"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_match_single_batch(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gzip
    import emu
    from tokenizer_amd import _native as N
    lib = emu.library()
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "gpt2.tiktoken.gz"), "rb").read())
    enc = N.Encoder(N.Vocab(raw, lib), N.CL100K)
    docs = [N.corpus_doc_host(1, 0x5EED0002, d, 20, 300, lib=lib) for d in range(301)]
    data = np.frombuffer(b"".join(docs), np.uint8)
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.int64)
    ids, ooffs = enc.encode_batch(data, offs)
    (r0, ids0, oo0, db0, tb0, nd0, nb0, nt0), (r1, ids1, oo1, db1, tb1, nd1, nb1, nt1) = res
    assert ids0 + ids1 == ids.tolist()
    assert (db0, tb0) == (0, 0) and db1 == 150 and tb1 == len(ids0)
    assert nd0 == nd1 == 301 and nb0 == nb1 == len(data) and nt0 == nt1 == len(ids)
    assert oo0 + [x + tb1 for x in oo1[1:]] == ooffs.tolist()
    # the shard files the two ranks wrote concatenate into the whole-batch result without any further exchange
    from tokenizer_amd import Shard
    shards = [Shard(str(tmp_path / ("tokens.%05d.tkzs" % r))) for r in range(2)]
    assert [sh.doc_base for sh in shards] == [0, 150] and [sh.token_base for sh in shards] == [0, len(ids0)]
    assert sum(len(sh) for sh in shards) == 301
    for sh in shards:
        for d in (0, len(sh) // 2, len(sh) - 1):
            g = sh.doc_base + d
            assert sh[d].tolist() == ids[ooffs[g]:ooffs[g + 1]].tolist()
    assert np.concatenate([np.asarray(sh.ids) for sh in shards]).tolist() == ids.tolist()


@pytest.mark.parametrize("world", [2, 8])
def test_tkz_comm_fake_rccl(tmp_path, world):
    """tkz_comm_* (the C ABI's direct-RCCL count exchange) at world sizes RCCL itself cannot run at here: the real tkz_comm.cpp in the
    CPU-emulated build, its dlopen("librccl.so.1") satisfied by tests/hostemu/fake_rccl.cpp (a Unix-socket all-gather) through LD_LIBRARY_PATH.
    Exercises the id exchange, ncclCommInitRank's argument order, the ncclInt64 value, count = 3, the table layout and tkz_shard_bases."""
    import json
    import subprocess
    import emu
    emu.library()
    fake = os.path.join(emu.EMU_DIR, "_build", "fake_rccl")
    assert os.path.exists(os.path.join(fake, "librccl.so.1"))
    env = dict(os.environ, LD_LIBRARY_PATH=fake + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_worker.py"), str(r), str(world), str(tmp_path / "id.bin"), str(tmp_path / ("out%d.json" % r))],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
    res = [json.load(open(tmp_path / ("out%d.json" % r))) for r in range(world)]
    table = [x for r in res for x in r["mine"]]
    doc_base = tok_base = byte_base = 0
    for r in res:
        assert r["world"] == world and r["comm_rank"] == r["rank"] and r["backend"].startswith("rccl 2.")
        assert r["table_dev"] == table and r["table_host"] == table              # every rank holds every rank's counts, in rank order
        assert r["bases"] == [doc_base, byte_base, tok_base]
        assert r["totals"] == [sum(table[0::3]), sum(table[1::3]), sum(table[2::3])] and r["totals"][0] == 173
        doc_base += r["mine"][0]; byte_base += r["mine"][1]; tok_base += r["mine"][2]
    # the shards concatenate into the whole batch
    import gzip
    from tokenizer_amd import _native as N
    lib = emu.library()
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "gpt2.tiktoken.gz"), "rb").read())
    docs = [N.corpus_doc_host(1, 0x5EED0002, d, 20, 200, lib=lib) for d in range(173)]
    ids, _ = N.Encoder(N.Vocab(raw, lib), N.CL100K).encode_batch(np.frombuffer(b"".join(docs), np.uint8), np.cumsum([0] + [len(d) for d in docs]).astype(np.int64))
    assert [x for r in res for x in r["ids"]] == ids.tolist()


def test_shard_file_round_trip_and_errors(tmp_path):
    from tokenizer_amd import Shard, write_shard
    ids = np.arange(17, dtype=np.int32) * 3
    offs = np.array([0, 0, 5, 5, 16, 17], dtype=np.int64)          # empty documents at both ends of a run
    p = str(tmp_path / "a.tkzs")
    write_shard(p, ids, offs, doc_base=1000, token_base=12345)
    sh = Shard(p)
    assert (len(sh), sh.n_tokens, sh.doc_base, sh.token_base) == (5, 17, 1000, 12345)
    assert [sh[d].tolist() for d in range(5)] == [[], [0, 3, 6, 9, 12], [], list(range(15, 48, 3)), [48]]
    write_shard(p, np.zeros(0, np.int32), np.zeros(1, np.int64))   # an empty shard is a valid file
    assert len(Shard(p)) == 0 and Shard(p).n_tokens == 0
    with pytest.raises(ValueError):
        write_shard(p, ids, np.array([0, 5, 3, 17], dtype=np.int64))
    with pytest.raises(ValueError):
        write_shard(p, ids, np.array([0, 16], dtype=np.int64))
    open(p, "wb").write(b"not a shard")
    with pytest.raises(ValueError):
        Shard(p)
