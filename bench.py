#!/usr/bin/env python3
"""bench.py -- input MB/s of the batch BPE encode path on MI355X (BASELINE.json's metric).

One "step" = one pass of the whole hot path (document marks, pre-tokenizer, piece lookup + byte-pair
merge, compaction, per-document offsets; plus, at N > 1, the all-gather of the per-rank counts) over
one batch of synthetic documents that is already resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[1] -- cl100k_base split pattern, 10 M synthetic ASCII
documents of 256..768 bytes (mean 512) per GPU, generated on the device by the counter-based generator
of tokenizer_amd/csrc/tkz_corpus.h (seed 0x5EED0002).  The cl100k_base rank file is downloaded by the
reference at run time and is not available offline: unless $TKZ_VOCAB_DIR/cl100k_base.tiktoken exists the
run uses the gpt2 rank file (the one vocabulary the reference ships) with the cl100k pattern and says so
(config.vocab = "gpt2 (VOCAB-SUBSTITUTED)").

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM bandwidth using the
algorithmic bytes of SURVEY.md 8(d); `cpu_baseline` times the reference-algorithm CPU restatement
(oracle/, "port") on a bounded sample of the same documents and is also the parity check of the run.
"""
import argparse
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md)


def load_vocab_bytes():
    d = os.environ.get("TKZ_VOCAB_DIR")
    if d and os.path.exists(os.path.join(d, "cl100k_base.tiktoken")):
        return open(os.path.join(d, "cl100k_base.tiktoken"), "rb").read(), "cl100k_base"
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "gpt2.tiktoken.gz"), "rb").read())
    return raw, "gpt2 (VOCAB-SUBSTITUTED: cl100k_base.tiktoken is not available offline)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=10_000_000, help="documents per GPU")
    ap.add_argument("--kind", type=int, default=1, help="corpus kind: 1 ASCII (config 2), 2 mixed UTF-8 (config 3), 3 long-context (config 5)")
    ap.add_argument("--min-len", type=int, default=256)
    ap.add_argument("--max-len", type=int, default=768)
    ap.add_argument("--pattern", type=int, default=2, help="1 pattern-1, 2 cl100k, 3 o200k")
    ap.add_argument("--cpu-sample-docs", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from tokenizer_amd import _native as N
    from tokenizer_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    raw, vocab_name = load_vocab_bytes()
    vocab = N.Vocab(raw)
    enc = N.Encoder(vocab, args.pattern, device=local_rank)

    # ---- synthetic corpus, generated on the device; rank r owns documents [r*docs, (r+1)*docs) ----
    seed = 0x5EED0000 + {1: 2, 2: 3, 3: 5}[args.kind]
    n_docs = args.docs
    first_doc = rank * n_docs
    d_offs = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    total = N.corpus_generate_device(local_rank, args.kind, seed, first_doc, n_docs, args.min_len, args.max_len,
                                     d_offs.data_ptr(), None, 0, stream)
    d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    N.corpus_generate_device(local_rank, args.kind, seed, first_doc, n_docs, args.min_len, args.max_len,
                             d_offs.data_ptr(), d_bytes.data_ptr(), total, stream)
    d_ids = torch.empty(total, dtype=torch.int32, device=dev)          # tokens <= bytes: always enough
    d_ooffs = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)

    def step():
        ntok = enc.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, total, d_ids.data_ptr(), total,
                                       d_ooffs.data_ptr(), stream)
        return sharded.gather_counts(n_docs, total, ntok, device=dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        g = step()
    enc.set_profiling(True)
    enc.kernel_ms(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = step()
    fence()
    dt = time.perf_counter() - t0
    enc.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kms = enc.kernel_ms()
    n_tokens_rank = int(g["table"][rank][2])

    if rank == 0:
        job_bytes, job_tokens, job_docs = g["bytes"], g["tokens"], g["docs"]
        ms_per_step = dt / args.steps * 1e3
        value = job_bytes * args.steps / dt / 1e6
        # ---- roofline of the dominant kernel (HBM-bound integer/indexing work; no MFMA) ----
        dom = max(kms, key=lambda k: kms[k][0])
        dom_ms = kms[dom][0] / max(1, kms[dom][1])
        alg_bytes = total + 4 * n_tokens_rank + 16 * n_docs      # read text + write int32 ids + 8 B offset in + 8 B offset out
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("docs_per_gpu") == n_docs and tj.get("kind") == args.kind:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_ms, 4),
                    "kernels_ms": {k: round(v[0] / max(1, v[1]), 4) for k, v in kms.items()}}
        # ---- CPU baseline (the oracle = reference-algorithm restatement, "port") + parity on the sample ----
        cpu = None
        host_path = None
        parity_note = "unchecked"
        if world == 1 and not args.no_cpu_baseline:
            try:
                from oracle import oracle as O
                ns = min(args.cpu_sample_docs, n_docs)
                h_offs = d_offs[:ns + 1].cpu().numpy()
                nb = int(h_offs[-1])
                h_bytes = d_bytes[:nb].cpu().numpy()
                h_ooffs = d_ooffs[:ns + 1].cpu().numpy()
                h_ids = d_ids[:int(h_ooffs[-1])].cpu().numpy()
                ov = O.Vocab(raw)
                threads = max(1, min(os.cpu_count() or 1, 64))
                tm = {}
                o_ids, o_counts = O.encode_batch(ov, args.pattern, h_bytes, h_offs, threads=threads, timing=tm)
                tcpu = tm["seconds"]
                same = len(o_ids) == len(h_ids) and np.array_equal(o_ids, h_ids) and np.array_equal(np.diff(h_ooffs), o_counts)
                # (not timed) the LAST documents of the batch too, and the offsets of the whole batch: placement at large indices
                nt = min(20_000, n_docs)
                t_offs = d_offs[n_docs - nt:n_docs + 1].cpu().numpy()
                t_bytes = d_bytes[int(t_offs[0]):int(t_offs[-1])].cpu().numpy()
                t_ooffs = d_ooffs[n_docs - nt:n_docs + 1].cpu().numpy()
                t_ids = d_ids[int(t_ooffs[0]):int(t_ooffs[-1])].cpu().numpy()
                p_ids, p_counts = O.encode_batch(ov, args.pattern, t_bytes, t_offs - t_offs[0], threads=threads)
                same = same and np.array_equal(p_ids, t_ids) and np.array_equal(np.diff(t_ooffs), p_counts)
                same = same and int(d_ooffs[0].item()) == 0 and int(d_ooffs[n_docs].item()) == n_tokens_rank \
                    and bool((d_ooffs[1:n_docs + 1] >= d_ooffs[:n_docs]).all().item())
                parity_note = ("bit-exact vs oracle on the first %d and the last %d docs (%d tokens); offsets of all %d docs monotone, ending at the token count"
                               % (ns, nt, len(o_ids) + len(p_ids), n_docs)) if same else "MISMATCH vs oracle on the sample"
                # one thread, on a tenth of the sample (SURVEY.md 8d asks for both figures)
                n1 = max(1, ns // 20)
                O.encode_batch(ov, args.pattern, h_bytes[:int(h_offs[n1])], h_offs[:n1 + 1], threads=1, timing=tm)
                t1 = tm["seconds"]
                cpu_1t = round(int(h_offs[n1]) / t1 / 1e6, 2)
                # PCIe-inclusive rate through the host-buffer entry point (tkz_encode_batch_utf8: H2D of the text, the kernels,
                # D2H of ids + offsets), on ordinary (pageable) numpy buffers and on page-locked ones.  Output buffers are
                # allocated and touched beforehand: a fresh np.empty would add its first-touch page faults to the figure.
                # Reported beside the number, never as `value`.
                try:
                    nh = min(1_000_000, n_docs)
                    hh_offs = d_offs[:nh + 1].cpu().numpy()
                    hh_bytes = d_bytes[:int(hh_offs[-1])].cpu().numpy()
                    o_ids_buf = np.zeros(len(hh_bytes), np.int32)
                    o_off_buf = np.zeros(nh + 1, np.int64)
                    rates = []
                    for pinned in (False, True):
                        if pinned:
                            tb = torch.empty(len(hh_bytes), dtype=torch.uint8).pin_memory(); tb.numpy()[:] = hh_bytes
                            to = torch.empty(nh + 1, dtype=torch.int64).pin_memory(); to.numpy()[:] = hh_offs
                            ti = torch.zeros(len(hh_bytes), dtype=torch.int32).pin_memory()
                            too = torch.zeros(nh + 1, dtype=torch.int64).pin_memory()
                            bufs = (tb.numpy(), to.numpy(), (ti.numpy(), too.numpy()))
                        else:
                            bufs = (hh_bytes, hh_offs, (o_ids_buf, o_off_buf))
                        enc.encode_batch(bufs[0], bufs[1], out=bufs[2])                   # sizes the encoder's staging buffers
                        tc = time.perf_counter()
                        r_ids, r_ooffs = enc.encode_batch(bufs[0], bufs[1], out=bufs[2])
                        rates.append(round(len(hh_bytes) / (time.perf_counter() - tc) / 1e6, 1))
                        host_same = (pinned is False or host_same) and int(r_ooffs[-1]) == int(d_ooffs[nh].item()) \
                            and np.array_equal(r_ids[:len(h_ids)], h_ids[:len(r_ids)])
                    host_path = {"value": rates[0], "value_pinned_buffers": rates[1], "unit": "MB/s", "docs": nh, "same_ids_as_device_path": bool(host_same),
                                 "note": "tkz_encode_batch_utf8 on host buffers: H2D of the text + kernels + D2H of ids and offsets, one after the other"}
                except Exception as ex:                      # an auxiliary figure must never cost the bench line
                    host_path = {"error": "%s: %s" % (type(ex).__name__, ex)}
                cpu = {"value": round(nb / tcpu / 1e6, 2), "unit": "MB/s", "cores": threads, "kind": "port", "value_1_thread": cpu_1t,
                       "sample": "first %d documents (%.1f MB) of the same corpus, reference-algorithm CPU restatement (oracle/), "
                                 "8192-entry LRU memo per thread, %d threads of %d host cores" % (ns, nb / 1e6, threads, os.cpu_count() or 1)}
            except Exception as ex:                          # (e.g. no C compiler for the oracle on this host)
                parity_note = "unchecked: CPU oracle unavailable (%s: %s)" % (type(ex).__name__, ex)
        line = {
            "metric": "input MB/s encoded (cl100k_base)", "value": round(value, 1), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: cl100k_base pattern, %d synthetic ASCII docs/GPU, %d..%d B (mean %.0f), device-resident"
                                   % (n_docs, args.min_len, args.max_len, job_bytes / max(1, job_docs)) if args.kind == 1 and args.pattern == 2 else
                                   "kind %d corpus, pattern %d, %d docs/GPU, %d..%d B" % (args.kind, args.pattern, n_docs, args.min_len, args.max_len),
                       "vocab": vocab_name, "docs_per_gpu": n_docs, "bytes_per_gpu": total, "tokens_per_gpu": n_tokens_rank,
                       "job_docs": job_docs, "job_bytes": job_bytes, "job_tokens": job_tokens,
                       "partitioning": "contiguous document ranges, one process per GPU; all-gather of 3 int64 counts per rank"},
            "tokens_per_s": round(job_tokens * args.steps / dt, 1),
            "parity": parity_note,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "pcie_inclusive": host_path,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
