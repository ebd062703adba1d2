import sys, time, gzip, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tokenizer_amd import _native as N
dev = torch.device("cuda", 0)
n = 512 << 20
h_page = torch.empty(n, dtype=torch.uint8); h_page.fill_(1)
h_pin = torch.empty(n, dtype=torch.uint8).pin_memory(); h_pin.fill_(1)
d = torch.empty(n, dtype=torch.uint8, device=dev)
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
print("H2D pageable GB/s", n / t(lambda: d.copy_(h_page)) / 1e9)
print("H2D pinned   GB/s", n / t(lambda: d.copy_(h_pin, non_blocking=True)) / 1e9)
print("D2H pageable GB/s", n / t(lambda: h_page.copy_(d)) / 1e9)
print("D2H pinned   GB/s", n / t(lambda: h_pin.copy_(d, non_blocking=True)) / 1e9)
raw = gzip.decompress(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/gpt2.tiktoken.gz"), "rb").read())
enc = N.Encoder(N.Vocab(raw), 2, device=0)
nd = 1_000_000
d_offs = torch.empty(nd + 1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
total = N.corpus_generate_device(0, 1, 0x5EED0002, 0, nd, 256, 768, d_offs.data_ptr(), None, 0, st)
d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
N.corpus_generate_device(0, 1, 0x5EED0002, 0, nd, 256, 768, d_offs.data_ptr(), d_bytes.data_ptr(), total, st)
hb = d_bytes[:total].cpu().numpy(); ho = d_offs.cpu().numpy()
for rep in range(3):
    t0 = time.perf_counter(); ids, oo = enc.encode_batch(hb, ho); dt = time.perf_counter() - t0
    print("pageable call", rep, "GB/s", total / dt / 1e9, "ms", dt * 1e3)
pb = torch.empty(total, dtype=torch.uint8).pin_memory(); pb.numpy()[:] = hb
po = torch.empty(nd + 1, dtype=torch.int64).pin_memory(); po.numpy()[:] = ho
pi = torch.empty(total, dtype=torch.int32).pin_memory(); poo = torch.empty(nd + 1, dtype=torch.int64).pin_memory()
for rep in range(3):
    t0 = time.perf_counter(); ids2, oo2 = enc.encode_batch(pb.numpy(), po.numpy(), out=(pi.numpy(), poo.numpy())); dt = time.perf_counter() - t0
    print("pinned call", rep, "GB/s", total / dt / 1e9, "ms", dt * 1e3)
print(np.array_equal(ids, ids2))
# UTF-16 batch entry: the same documents as code units (ASCII corpus: one unit per byte)
hu = hb.astype(np.uint16)
obuf = np.zeros(total, np.int32); oobuf = np.zeros(nd + 1, np.int64)
for rep in range(3):
    t0 = time.perf_counter(); ids3, oo3 = enc.encode_batch_utf16(hu, ho, out=(obuf, oobuf)); dt = time.perf_counter() - t0
    print("utf16 call", rep, "GB/s of UTF-8-equivalent text", total / dt / 1e9, "ms", dt * 1e3)
print(np.array_equal(ids, ids3), np.array_equal(oo, oo3))
