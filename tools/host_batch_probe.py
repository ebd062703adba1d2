"""One tkz_encode_batch_utf8 call on host buffers of 16 MB .. 512 MB (documents of ~512 bytes, the bench corpus), page-locked and pageable, outputs allocated and
touched beforehand: MB/s of text, the best of `reps` calls.  For A/Bs of the chunk pipeline (TKZ_LIBTKZ names the library; tools/gpu_job_sdma.sh)."""
import sys, time, gzip, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tokenizer_amd import _native as N
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
raw = gzip.decompress(open(os.path.join(root, "tests/golden/gpt2.tiktoken.gz"), "rb").read())
enc = N.Encoder(N.Vocab(raw), 2, device=0)
dev = torch.device("cuda", 0)
sizes = [float(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["16", "64", "256", "512"])]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
out = {}
st = torch.cuda.current_stream().cuda_stream
for mb in sizes:
    nd = int(mb * (1 << 20) / 512)
    d_offs = torch.empty(nd + 1, dtype=torch.int64, device=dev)
    total = N.corpus_generate_device(0, 1, 0x5EED0002, 0, nd, 256, 768, d_offs.data_ptr(), None, 0, st)
    d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    N.corpus_generate_device(0, 1, 0x5EED0002, 0, nd, 256, 768, d_offs.data_ptr(), d_bytes.data_ptr(), total, st)
    hb = d_bytes[:total].cpu().numpy(); ho = d_offs.cpu().numpy()
    row = {"bytes": int(total)}
    ref = None
    for pinned in (True, False):
        if pinned:
            tb = torch.empty(total, dtype=torch.uint8).pin_memory(); tb.numpy()[:] = hb
            to = torch.empty(nd + 1, dtype=torch.int64).pin_memory(); to.numpy()[:] = ho
            ti = torch.zeros(total, dtype=torch.int32).pin_memory(); too = torch.zeros(nd + 1, dtype=torch.int64).pin_memory()
            bufs = (tb.numpy(), to.numpy(), (ti.numpy(), too.numpy()))
        else:
            bufs = (hb, ho, (np.zeros(total, np.int32), np.zeros(nd + 1, np.int64)))
        enc.encode_batch(bufs[0], bufs[1], out=bufs[2])
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter(); ids, oo = enc.encode_batch(bufs[0], bufs[1], out=bufs[2]); best = min(best, time.perf_counter() - t0)
        key = "pinned" if pinned else "pageable"
        row[key + "_ms"] = round(best * 1e3, 3); row[key + "_MBps"] = round(total / best / 1e6, 1)
        if ref is None: ref = (ids.copy(), oo.copy()); row["tokens"] = int(oo[-1])
        else: row["same_ids"] = bool(np.array_equal(ref[0], ids) and np.array_equal(ref[1], oo))
    out["%g MB" % mb] = row
print(json.dumps(out))
