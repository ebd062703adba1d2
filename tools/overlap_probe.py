"""Does an HBM-bound kernel on a second stream overlap with the (latency-bound) encode kernels?  Development probe."""
import sys, time, gzip, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tokenizer_amd import _native as N
dev = torch.device("cuda", 0)
raw = gzip.decompress(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/gpt2.tiktoken.gz"), "rb").read())
enc = N.Encoder(N.Vocab(raw), 2, device=0)
nd = 10_000_000
d_offs = torch.empty(nd + 1, dtype=torch.int64, device=dev)
sA = torch.cuda.Stream(); sB = torch.cuda.Stream()
st = sA.cuda_stream
total = N.corpus_generate_device(0, 1, 0x5EED0002, 0, nd, 256, 768, d_offs.data_ptr(), None, 0, st)
d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
N.corpus_generate_device(0, 1, 0x5EED0002, 0, nd, 256, 768, d_offs.data_ptr(), d_bytes.data_ptr(), total, st)
d_ids = torch.empty(total, dtype=torch.int32, device=dev); d_oo = torch.empty(nd + 1, dtype=torch.int64, device=dev)
x = torch.zeros(750_000_000, dtype=torch.int32, device=dev)      # 3 GB: one add_ = 6 GB of traffic
def encode():
    enc.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), nd, total, d_ids.data_ptr(), total, d_oo.data_ptr(), st)
def hbm(n):
    with torch.cuda.stream(sB):
        for _ in range(n): x.add_(1)
for _ in range(2): encode(); hbm(2); torch.cuda.synchronize()
t0 = time.perf_counter(); encode(); torch.cuda.synchronize(); te = time.perf_counter() - t0
t0 = time.perf_counter(); hbm(10); torch.cuda.synchronize(); tb = time.perf_counter() - t0
t0 = time.perf_counter(); hbm(10); encode(); torch.cuda.synchronize(); tc = time.perf_counter() - t0
print("encode %.1f ms   10 x add_ (60 GB) %.1f ms   together %.1f ms   (sum %.1f)" % (te * 1e3, tb * 1e3, tc * 1e3, (te + tb) * 1e3))
