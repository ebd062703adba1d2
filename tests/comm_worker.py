"""One rank of tests/test_sharded.py::test_tkz_comm_fake_rccl: the REAL tkz_comm.cpp (CPU-emulated build) over tests/hostemu/fake_rccl.cpp.
No torch in this process -- torch bundles a librccl.so.1 of its own, and tkz_comm.cpp's dlopen would be handed that copy.
usage: comm_worker.py rank world id_file out_file"""
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, id_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    assert "torch" not in sys.modules
    import ctypes as C
    import emu
    from tokenizer_amd import _native as N
    from tokenizer_amd import sharded
    lib = emu.library()
    assert lib.has_comm

    def exchange(idbytes):          # the 128-byte id travels through a file
        if rank == 0:
            with open(id_file + ".tmp", "wb") as f:
                f.write(idbytes)
            os.replace(id_file + ".tmp", id_file)
            return idbytes
        for _ in range(3000):
            if os.path.exists(id_file):
                return open(id_file, "rb").read()
            time.sleep(0.01)
        raise RuntimeError("no id file")

    comm = N.Comm(rank, world, 0, exchange, lib=lib)
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "gpt2.tiktoken.gz"), "rb").read())
    enc = N.Encoder(N.Vocab(raw, lib), N.CL100K)
    n_total = 173
    lo, hi = sharded.shard_range(n_total, rank, world, lib=lib)
    docs = [N.corpus_doc_host(1, 0x5EED0002, d, 20, 200, lib=lib) for d in range(lo, hi)]
    data = np.frombuffer(b"".join(docs), np.uint8) if docs else np.zeros(0, np.uint8)
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.int64)
    ids, ooffs = enc.encode_batch(data, offs)
    # the device form: the encoder's own count block (written by the batch, the single-launch path included) straight into ncclAllGather
    table_dev = np.full(world * 3, -1, np.int64)
    comm.allgather_counts_device(enc.counts_device, table_dev.ctypes.data)
    # the host form
    table_host = comm.allgather_counts(hi - lo, int(offs[-1]), len(ids))
    bases, totals = N.shard_bases(table_host, rank, lib=lib)
    json.dump({"rank": rank, "world": comm.world, "comm_rank": comm.rank, "backend": comm.backend, "mine": [hi - lo, int(offs[-1]), len(ids)],
               "table_dev": table_dev.tolist(), "table_host": table_host.reshape(-1).tolist(), "bases": [int(x) for x in bases],
               "totals": [int(x) for x in totals], "ids": ids.tolist()}, open(out_file, "w"))
    comm.close()


if __name__ == "__main__":
    main()
