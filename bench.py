#!/usr/bin/env python3
"""bench.py -- input MB/s of the batch BPE encode path on MI355X (BASELINE.json's metric).

One "step" = one pass of the whole hot path (document marks, pre-tokenizer, piece lookup + byte-pair merge, compaction,
per-document offsets, and the all-gather of the per-rank counts on the C ABI's RCCL communicator) over one batch of synthetic
documents that is already resident in HBM.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script re-launches itself under torch.distributed.run (one process per
GPU, rendezvous on 127.0.0.1); launched by the driver under torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.

Workload (config.workload): BASELINE.json configs[1] -- cl100k_base split pattern, 10 M synthetic ASCII documents of 256..768
bytes (mean 512) per GPU, generated on the device by the counter-based generator of tokenizer_amd/csrc/tkz_corpus.h (seed
0x5EED0002; 4096-word Zipf table, identifiers, URLs, hex blobs, numbers, contractions).  The cl100k_base rank file is
downloaded by the reference at run time and exists nowhere offline: unless $TKZ_VOCAB_DIR/cl100k_base.tiktoken is supplied the
run uses tests/golden/synth100k.tiktoken.gz, a trained stand-in of cl100k_base's size (100,256 keys, tools/train_bpe.py), and
says so in config.vocab.

The piece memo (the reference's LRUCache on the device) persists from call to call like the reference's: the W warm-up steps encode
OTHER documents of the same generator, then one untimed pass with the memo switched off sizes the workspace for the bench batch;
--no-memo measures without it (config.piece_memo says which).  Companion figures of the same line, never `value`: `value_no_memo`
(the timed steps again with the memo off) and `value_two_in_flight` (the same batch two at a time through
tkz_encode_batch_device_begin / _end on two streams: what keeping batches in flight buys over one synchronous call after the other).

Further companions: `piece_stats` (what the timed steps met: pieces, whole-piece hit rate, misses by kind, memo lookups and hits),
`value_heldout_vocab` (the same corpus under synth100k_heldout, a stand-in of the same size trained WITHOUT the bench's generator: synth100k
has seen the generator's output, the real cl100k_base lies between the two), `pcie_inclusive` (the host-buffer entry point on numpy buffers)
and `value_host_api` (tkz::TikTokenizer::EncodeBatchFlat on a million std::strings: the ITokenizer-shaped surface, gather included).

Rank 0 prints ONE JSON line.  `roofline` prices the step against HBM bandwidth with the algorithmic bytes of SURVEY.md 8(d) (`frac`: the
whole launch sequence; `frac_dominant`: the dominant kernel alone; `traffic_pipeline` / `wasted`: the counted HBM bytes of all kernels and
their ratio to the algorithmic ones); `cpu_baseline` times the reference-algorithm CPU restatement (oracle/, kind "port") on a bounded sample of the
same documents on ALL host cores and is also the parity check of the run.
"""
import argparse
import glob
import gzip
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md)
VALU_PEAK_TPS = 256 * 4 * 2.4e9 / 4 / 1e12      # wave64 VALU instructions a second, in T/s: 256 CUs x 4 SIMDs, one instruction per 4 cycles at 2.4 GHz
VOCAB_OF_PATTERN = {1: ("gpt2", None), 2: ("synth100k", "cl100k_base"), 3: ("synth200k", "o200k_base"), 4: ("synth200k", "o200k_base")}
PATTERN_NAME = {1: "pattern 1 (gpt2 / r50k / p50k)", 2: "cl100k_base", 3: "o200k_base (ECMAScript reading: the TypeScript reference's engine)",
                4: "o200k_base (.NET reading: the string through the C# reference's Regex)"}


def load_vocab_bytes(pattern, want=None):
    """(bytes, label).  The real rank file from $TKZ_VOCAB_DIR when present; otherwise the stand-in of the same size."""
    stand_in, real = VOCAB_OF_PATTERN[pattern]
    if want:
        stand_in, real = want, None
    d = os.environ.get("TKZ_VOCAB_DIR")
    if real and d and os.path.exists(os.path.join(d, real + ".tiktoken")):
        return open(os.path.join(d, real + ".tiktoken"), "rb").read(), real + " (REAL-VOCAB: $TKZ_VOCAB_DIR/" + real + ".tiktoken)"
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", stand_in + ".tiktoken.gz"), "rb").read())
    if stand_in == "gpt2":
        return raw, "gpt2" + ("" if pattern == 1 else " (VOCAB-SUBSTITUTED)")
    return raw, "%s (VOCAB-SUBSTITUTED: trained stand-in with the key count of %s, which is not available offline)" % (stand_in, real or "the named vocabulary")


def kernel_sources_sha():
    """Identity of the kernels a PMC traffic figure was collected on: sha256 over tokenizer_amd/csrc (there is no .git on the GPU box)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "tokenizer_amd", "csrc", "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def relaunch_under_torchrun(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def reference_dotnet_baseline(sample_path):
    """when a .NET SDK and a checkout of the reference are present on this host, time the REAL TokenizerLib
    through tools/dotnet_baseline (a console driver).  Neither exists in this image; the probe is the hook."""
    ref = os.environ.get("TKZ_REFERENCE_DIR")
    if not shutil.which("dotnet") or not ref or not os.path.isdir(os.path.join(ref, "Tokenizer_C#", "TokenizerLib")):
        return None
    try:
        proj = os.path.join(ROOT, "tools", "dotnet_baseline")
        out = subprocess.run(["dotnet", "run", "-c", "Release", "--project", proj, "--", sample_path], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, TKZ_REFERENCE_DIR=ref))
        for line in out.stdout.splitlines():
            if line.startswith("{"):
                return json.loads(line)
        return {"error": (out.stderr or out.stdout)[-400:]}
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}


# ---- real text (kind 6) -------------------------------------------------------------------------------------------------------------------
# The reference benches real words (PerfBenchmark/Program.cs:14-38), the synthetic kinds a 4096-word Zipf lexicon.  Kind 6 is the source and
# documentation text that exists on the box -- the Python standard library, the installed Python packages (torch and its headers among them),
# the ROCm and system C / C++ headers --, every file once, in sorted order, NOT tiled, cut into documents of min..max bytes at character
# boundaries.  Files that are not well-formed UTF-8 or hold a NUL byte are left out.  The sha256 of the concatenation goes into the bench line.
REAL_TEXT_ROOTS = ["/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include", "/usr/include"]
REAL_TEXT_EXT = {".py", ".pyi", ".h", ".hpp", ".c", ".cc", ".cpp", ".cu", ".cuh", ".hip", ".md", ".rst", ".js", ".html", ".cmake", ".css", ".sh", ".toml", ".cfg"}


def real_text_corpus(limit_bytes, min_len, max_len, seed=0x5EED0006, roots=None):
    """(bytes uint8[total], offsets int64[n + 1], meta).  limit_bytes <= 0: everything there is."""
    import numpy as np
    t0 = time.perf_counter()
    files = []
    for r in (roots or REAL_TEXT_ROOTS):
        for dp, dn, fn in os.walk(r):
            dn.sort()
            for f in sorted(fn):
                if os.path.splitext(f)[1].lower() in REAL_TEXT_EXT:
                    q = os.path.join(dp, f)
                    if not os.path.islink(q):
                        files.append(q)
    parts, total, used, skipped = [], 0, 0, 0
    h = hashlib.sha256()
    for q in files:
        if limit_bytes > 0 and total >= limit_bytes:
            break
        try:
            b = open(q, "rb").read()
            if b"\0" in b:
                raise ValueError
            b.decode("utf-8")
        except (OSError, ValueError):
            skipped += 1
            continue
        if not b:
            continue
        parts.append(b)
        h.update(b)
        total += len(b)
        used += 1
    data = np.frombuffer(b"".join(parts), np.uint8) if parts else np.zeros(0, np.uint8)
    del parts
    # document cuts: lengths drawn uniformly from min..max, every cut moved forward to the next character boundary
    rng = np.random.default_rng(seed)
    n_guess = int(total // max(1, (min_len + max_len) // 2) * 1.05) + 4096
    cuts = np.cumsum(rng.integers(min_len, max_len + 1, n_guess, dtype=np.int64))
    cuts = cuts[cuts < total]
    if len(cuts):
        lead = (data & 0xC0) != 0x80                              # True at the first byte of a character
        nxt = np.flatnonzero(lead)
        cuts = nxt[np.minimum(np.searchsorted(nxt, cuts), len(nxt) - 1)]      # first lead byte at or after the cut (the text ends in a whole char)
        cuts = np.unique(cuts[(cuts > 0) & (cuts < total)])
    offs = np.concatenate([[0], cuts, [total]]).astype(np.int64) if total else np.zeros(1, np.int64)
    meta = {"files": used, "files_skipped": skipped, "bytes": int(total), "docs": int(len(offs) - 1), "sha256": h.hexdigest(),
            "roots": roots or REAL_TEXT_ROOTS, "seconds_to_read": round(time.perf_counter() - t0, 2),
            "non_ascii_bytes": int((data >= 0x80).sum()) if total else 0}
    return data, offs, meta


# ---- the oracle as the checker on document ranges of a device-resident batch ----------------------------------------------------------------
def sample_ranges(n_docs, sample):
    """The first and the last `sample` documents of a shard (all of it when it has no more than two samples' worth)."""
    if sample <= 0 or n_docs <= 2 * sample:
        return [(0, n_docs)]
    return [(0, sample), (n_docs - sample, n_docs)]


def check_ranges(O, ov, pattern, d_bytes, d_offs, d_ids, d_ooffs, n_tokens, ranges, threads):
    """tkzo_check_batch on the documents [a, b) of every range: only those documents' bytes and ids are downloaded.  Returns
    {"docs", "bad", "first_bad" (document index in the shard or -1), "tokens", "bytes", "seconds"}."""
    import numpy as np
    res = {"docs": 0, "bad": 0, "first_bad": -1, "tokens": 0, "bytes": 0, "seconds": 0.0}
    for a, b in ranges:
        if b <= a:
            continue
        offs = d_offs[a:b + 1].cpu().numpy()
        oo = d_ooffs[a:b + 1].cpu().numpy()
        b0, b1, t0, t1 = int(offs[0]), int(offs[-1]), int(oo[0]), int(oo[-1])
        res["docs"] += b - a
        res["bytes"] += b1 - b0
        sane = 0 <= t0 <= t1 <= n_tokens and bool((np.diff(oo) >= 0).all()) and (t1 - t0) <= (b1 - b0)
        if not sane:                                             # (offsets that are not even monotone: every document of the range counts as wrong)
            res["bad"] += b - a
            res["first_bad"] = a if res["first_bad"] < 0 else res["first_bad"]
            continue
        tm = {}
        bad, first, otok = O.check_batch(ov, pattern, d_bytes[b0:b1].cpu().numpy(), offs - b0, d_ids[t0:t1].cpu().numpy(), oo - t0, threads=threads, timing=tm)
        if otok != t1 - t0 and bad == 0:
            bad, first = 1, 0
        res["bad"] += bad
        if bad and res["first_bad"] < 0:
            res["first_bad"] = a + max(0, first)
        res["tokens"] += otok
        res["seconds"] += tm["seconds"]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=None, help="documents per GPU (default: 10 M at --gpus 1 = BASELINE configs[1]; 12.5 M at --gpus > 1 = "
                                                           "one GPU's share of configs[3], 100 M documents over 8 GPUs)")
    ap.add_argument("--kind", type=int, default=1, help="corpus: 1 ASCII (config 2), 2 mixed UTF-8 (config 3), 3 long-context (config 5), "
                                                        "4 the reference's test text lib.rs.txt tiled (real source code), "
                                                        "5 ONE document of shuffled words joined by single spaces (the reference's own benchmark, PerfBenchmark/Program.cs:14-32), "
                                                        "6 REAL text: the source / documentation files on the box, every file once, not tiled (--real-text-mb)")
    ap.add_argument("--real-text-mb", type=int, default=None, help="kind 6: at most this many MB of the box's text (0: everything there is); "
                                                                   "default run (kind 1, N = 1): size of the `real_text` companion leg (default 256; 0: skip the leg)")
    ap.add_argument("--parity-sample-docs", type=int, default=200_000, help="N > 1: every rank checks the first and the last this-many documents of ITS OWN shard against the oracle")
    ap.add_argument("--emulated", action="store_true", help="TEST INFRASTRUCTURE (tests/test_bench_line.py): run the kernels through the CPU emulator of tests/hostemu with gloo "
                                                            "instead of a GPU and RCCL, to exercise this script's own logic at world sizes the 1-GPU boxes cannot run.  The line says so "
                                                            "(`emulated`: true, `data`: 'EMULATED ...') and is not a measurement of anything")
    ap.add_argument("--min-len", type=int, default=256)
    ap.add_argument("--max-len", type=int, default=768)
    ap.add_argument("--pattern", type=int, default=2, help="1 pattern-1, 2 cl100k, 3 o200k as the TypeScript reference's engine reads it, 4 o200k as .NET's Regex reads it")
    ap.add_argument("--vocab", default=None, help="gpt2 | synth100k | synth100k_heldout | synth200k (default: the stand-in of the pattern's vocabulary)")
    ap.add_argument("--parity-only", action="store_true", help="the oracle compares every document of the batch (all host cores) but the thread sweep of the CPU baseline, "
                                                               "the PCIe-inclusive and the host-API legs are skipped: for the shape runs of tools/gpu_job_*.sh")
    ap.add_argument("--heldout-steps", type=int, default=None, help="timed steps of the value_heldout_vocab leg (the same corpus under synth100k_heldout, a stand-in trained "
                                                                    "WITHOUT the bench's generator); default: as --steps at N = 1 with the cl100k pattern and the default vocabulary, else 0")
    ap.add_argument("--cpu-sample-docs", type=int, default=2_000_000, help="documents of the CPU-baseline thread sweep (the all-core run covers the whole batch)")
    ap.add_argument("--no-memo-steps", type=int, default=None, help="timed steps with the piece memo off (value_no_memo); default: as --steps")
    ap.add_argument("--pipelined-steps", type=int, default=4, help="timed steps of the two-batches-in-flight leg (value_two_in_flight; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-first-call", action="store_true", help="skip the `first_call` leg (a fresh encoder, reserved, the bench batch once): the profiling runs, whose per-launch averages are over the headline steps")
    ap.add_argument("--no-piece-stats", action="store_true", help="skip the untimed step that counts pieces / misses / memo hits (tools/gpu_profile.sh: only the headline steps, their "
                                                                  "warm-up and the sizing pass run under the profiler)")
    ap.add_argument("--no-memo", action="store_true", help="switch the piece memo (the device form of the reference's LRUCache) off")
    ap.add_argument("--write-shards", default=None, metavar="DIR", help="after the timed loop every rank writes its token shard file (SURVEY 8f-2)")
    args = ap.parse_args()

    if args.docs is None:
        args.docs = 10_000_000 if args.gpus == 1 else 12_500_000
        if args.kind == 5:
            args.docs = 131_072          # x 512 B = one 64 MB document
            args.min_len = args.max_len = 512
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)

    # stdout carries ONE JSON line and nothing else: libraries that print on file descriptor 1 (RCCL writes a version banner
    # through C stdio when its first communicator is created) are pointed at stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    emu = args.emulated
    if emu:
        # test infrastructure: the real kernel sources on the CPU emulator (tests/hostemu), "device" memory = host memory, gloo for RCCL
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu as emu_mod
        N._default = emu_mod.library()
        local_rank = 0
        dev = torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
    else:
        if torch.cuda.device_count() <= local_rank:
            print("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()), file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=dev)
    device_sync = (lambda: None) if emu else torch.cuda.synchronize

    raw, vocab_name = load_vocab_bytes(args.pattern, args.vocab)
    vocab = N.Vocab(raw)
    enc = N.Encoder(vocab, args.pattern, device=local_rank)

    # ---- the count exchange: the C ABI's own RCCL communicator (also at N = 1: RCCL is initialised and used on every run) ----
    def exchange(idbytes):
        if world == 1:
            return idbytes
        box = [idbytes]
        dist.broadcast_object_list(box, src=0, device=dev)
        return box[0]
    comm, comm_info = None, None
    if emu:
        comm_info = {"backend": "gloo (EMULATED run: torch.distributed all_gather of the counts, tokenizer_amd.sharded.gather_counts)", "world_size": world, "rank": rank}
    else:
        try:
            comm = sharded.RcclCounts(rank, world, local_rank, exchange)
            comm_info = comm.info()
        except Exception as ex:
            if world > 1:
                raise
            comm_info = {"error": "%s: %s" % (type(ex).__name__, ex)}

    # ---- synthetic corpus, generated on the device; rank r owns documents [r*docs, (r+1)*docs) ----
    n_docs = args.docs
    first_doc = rank * n_docs
    stream = 0 if emu else torch.cuda.current_stream().cuda_stream
    real_meta = None
    if args.kind == 6:
        # REAL text, every file once: rank r takes the r-th of `world` contiguous shares of the documents
        lim = 0 if args.real_text_mb is None else args.real_text_mb
        r_bytes, r_offs, real_meta = real_text_corpus(lim << 20, args.min_len, args.max_len)
        nd_all = len(r_offs) - 1
        lo, hi = N.shard_range(nd_all, rank, world)
        b0, b1 = int(r_offs[lo]), int(r_offs[hi])
        n_docs = hi - lo
        first_doc = lo
        total = b1 - b0
        d_bytes = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
        d_bytes[:total] = torch.from_numpy(r_bytes[b0:b1].copy()).to(dev)
        d_offs = torch.from_numpy((r_offs[lo:hi + 1] - b0).copy()).to(dev)
        real_meta["docs_this_rank"] = n_docs
        del r_bytes, r_offs
        seed = None
    elif args.kind == 4:
        # the reference's own test input (Tokenizer_C#/TokenizerTest/testData/lib.rs.txt, real Rust source) tiled; document lengths
        # drawn like the synthetic kinds.  A document is a slice of the tiled text, so pieces are cut at document edges as anywhere.
        text = torch.from_numpy(np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "lib.rs.txt"), "rb").read(), np.uint8).copy()).to(dev)
        g = torch.Generator(device=dev)
        g.manual_seed(0x5EED0004 + first_doc)
        lens = torch.randint(args.min_len, args.max_len + 1, (n_docs,), generator=g, device=dev, dtype=torch.int64)
        d_offs = torch.zeros(n_docs + 1, dtype=torch.int64, device=dev)
        d_offs[1:] = torch.cumsum(lens, 0)
        total = int(d_offs[-1].item())
        reps = (total + 64 + len(text) - 1) // len(text)
        d_bytes = text.repeat(reps)[:total + 64].contiguous()
        seed = None
    else:
        seed = 0x5EED0000 + {1: 2, 2: 3, 3: 5, 5: 6}[args.kind]
        d_offs = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
        total = N.corpus_generate_device(local_rank, args.kind, seed, first_doc, n_docs, args.min_len, args.max_len,
                                         d_offs.data_ptr(), None, 0, stream)
        d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
        N.corpus_generate_device(local_rank, args.kind, seed, first_doc, n_docs, args.min_len, args.max_len,
                                 d_offs.data_ptr(), d_bytes.data_ptr(), total, stream)
    gen_docs = n_docs
    if args.kind == 5:
        # the reference's own benchmark shape: ONE string of shuffled words joined by single spaces, one Encode call.  The generated
        # documents (words behind single spaces) stand back to back in d_bytes; the batch is that text as a single document.
        n_docs = 1
        d_offs = torch.tensor([0, total], dtype=torch.int64, device=dev)
    d_ids = torch.empty(total, dtype=torch.int32, device=dev)          # tokens <= bytes: always enough
    d_ooffs = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)

    # The piece memo is part of the hot path (the reference's LRUCache, TikTokenizer.cs:254,270) and persists from call to call, like the
    # reference's.  So that the timed steps do not meet a memo filled by THEMSELVES, the warm-up steps encode OTHER documents of the same
    # generator (the range behind every rank's own); the timed steps then run on the bench batch with the memo as those left it.
    memo_note = "off"
    warm = None
    if args.no_memo:
        enc.set_option(N.OPT_PIECE_MEMO, 0)
    elif seed is not None:
        w_first = (world + rank) * gen_docs
        w_offs = torch.empty(gen_docs + 1, dtype=torch.int64, device=dev)
        w_total = N.corpus_generate_device(local_rank, args.kind, seed, w_first, gen_docs, args.min_len, args.max_len, w_offs.data_ptr(), None, 0, stream)
        w_bytes = torch.empty(w_total + 64, dtype=torch.uint8, device=dev)
        N.corpus_generate_device(local_rank, args.kind, seed, w_first, gen_docs, args.min_len, args.max_len, w_offs.data_ptr(), w_bytes.data_ptr(), w_total, stream)
        w_ids = torch.empty(w_total, dtype=torch.int32, device=dev)
        w_docs = gen_docs
        if args.kind == 5:
            w_docs, w_offs = 1, torch.tensor([0, w_total], dtype=torch.int64, device=dev)
        warm = (w_bytes, w_offs, w_total, w_ids, w_docs)
        memo_note = "on: %d slots x %d-way buckets, filled during the warm-up steps from %d OTHER documents of the same generator (documents %d..)" % (
            enc.memo_slots, enc.memo_ways, gen_docs, w_first)
    elif args.kind == 6:
        # real, non-repeating text has no "other documents of the same generator": every timed step starts on an EMPTY memo, as a fresh
        # TikTokenizer's LRUCache does on a text it has never seen -- the memo then holds only what this very pass put there
        memo_note = "on: %d slots, EMPTIED before every step (real text is not tiled and there is no second corpus to warm it on: every step is a first pass)" % enc.memo_slots
    else:
        memo_note = "on: %d slots, filled during the warm-up steps from the same tiled text (every piece of it repeats)" % enc.memo_slots
    empty_memo_each_step = args.kind == 6 and not args.no_memo
    if empty_memo_each_step:
        enc.set_option(N.OPT_PROMOTE, 0)                          # (a first pass has nothing promoted: promoted pieces are memo answers of EARLIER text)
    memo_on = {}                                                    # encoder -> the memo is switched on (so that a step only empties a memo that is in use)

    def step(en=None):
        en = en or enc
        if empty_memo_each_step and memo_on.get(id(en), True):
            en.set_option(N.OPT_PIECE_MEMO, 2)                      # (on and emptied: a device synchronisation + a 16 MB memset, inside the timed region)
        ntok = en.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, total, d_ids.data_ptr(), total,
                                      d_ooffs.data_ptr(), stream)
        if comm is not None:
            comm.gather_async(en, stream)      # ncclAllGather of {docs, bytes, tokens}, enqueued on the encode stream; the table stays in HBM
        return ntok

    def fence():
        if world > 1:
            dist.barrier()
        device_sync()

    def prepare(en):
        """The W untimed warm-up steps of an encoder (on the OTHER documents when there are any: they fill the piece memo), then one untimed
        pass over the bench batch itself with the memo switched off (it neither reads nor fills it then): the workspace takes the size THIS
        batch needs -- record capacity, scratch of its giant pieces -- outside the timed region."""
        nt = 0
        for _ in range(args.warmup):
            if warm is not None:
                en.encode_batch_device(warm[0].data_ptr(), warm[1].data_ptr(), warm[4], warm[2], warm[3].data_ptr(), warm[2], d_ooffs.data_ptr(), stream)
            else:
                nt = step(en)
            en.piece_stats()       # (waits for the promotion a learning warm-up step left running behind it: the next warm-up step can be the second round)
        if warm is not None:       # (after 5 GB of other documents every memo slot is taken: the timed steps cannot add entries of their own)
            en.set_option(N.OPT_PIECE_MEMO, 0)
            nt = step(en)
            en.set_option(N.OPT_PIECE_MEMO, 1)
        # What the encoder promoted out of its memo during the warm-up stays; nothing more is learnt from here on: a timed step must not be the
        # encoder's next learning batch (it counts hits and ends with the key tables rebuilt on the host: tens of milliseconds, once or twice in an
        # encoder's life -- with --warmup 2 the second of them fell into the timed loop: 29.6 ms a step instead of 21.6)
        if not empty_memo_each_step:
            en.set_option(N.OPT_PROMOTE, 0)
        return nt

    def timed(en, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            nt = step(en)
        fence()
        d = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d, nt

    def stats_of(en):
        """One more untimed step with the counting switched on (TKZ_OPT_PIECE_STATS): what the timed steps met -- pieces, whole-piece hits,
        misses by kind, memo lookups and hits (the memo as the timed steps found it: full, so this step adds nothing to it)."""
        try:
            if empty_memo_each_step and memo_on.get(id(en), True):
                en.set_option(N.OPT_PIECE_MEMO, 2)                  # (the statistics of a FIRST pass, like the timed steps)
            en.set_option(N.OPT_PIECE_STATS, 1)
            en.piece_stats(reset=True)
            en.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, total, d_ids.data_ptr(), total, d_ooffs.data_ptr(), stream)   # (no collective: rank 0 alone)
            device_sync()
            st = en.piece_stats(reset=True)
            en.set_option(N.OPT_PIECE_STATS, 0)
            st.pop("batches", None)
            return st
        except Exception as ex:
            return {"error": "%s: %s" % (type(ex).__name__, ex)}

    ntok = prepare(enc)
    enc.set_profiling(True)
    enc.kernel_ms(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ntok = step()
    fence()
    dt = time.perf_counter() - t0
    dt_own = dt
    enc.set_profiling(False)
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        allt = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([dt], dtype=torch.float64, device=dev))
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in allt]
        dt = max(float(x.item()) for x in allt)              # the slowest rank's time is the job's
    kms = enc.kernel_ms()
    try:
        npromo = enc.piece_stats().get("promoted_pieces_in_tables", 0)
        promoted_note = ("%d memo entries promoted into the key tables during the warm-up steps (TKZ_OPT_PROMOTE, automatic: learnt from the warm-up documents, "
                         "not from the timed batch)" % npromo) if npromo else "none"
    except Exception as ex:
        promoted_note = "unknown (%s)" % ex
    kms_rank = rank
    if world > 1:
        # the roofline prices the SLOWEST rank's kernels (its time is the job's): every rank's per-kernel milliseconds travel with its step time
        names = sorted(kms)
        mine = torch.tensor([dt_own] + [kms[k][0] for k in names] + [float(kms[k][1]) for k in names], dtype=torch.float64, device=dev)
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        rows = [r.cpu().tolist() for r in rows]
        kms_rank = max(range(world), key=lambda r: rows[r][0])
        kms = {k: (rows[kms_rank][1 + i], int(rows[kms_rank][1 + len(names) + i])) for i, k in enumerate(names)}
    piece_stats = stats_of(enc) if (rank == 0 and not args.no_piece_stats) else None
    if world > 1:
        fence()
    # kind 6 only: the same steps WITHOUT emptying the memo in between (it holds what the passes before put there from this very text: an upper
    # bound, every piece has been seen) -- `value_warm_memo`
    dt_warm = None
    if empty_memo_each_step and args.steps > 0:
        memo_on[id(enc)] = False                                    # (step() leaves the memo alone)
        enc.set_option(N.OPT_PROMOTE, 1)
        for _ in range(min(6, 2 + int((1 << 30) // max(1, total)))):         # (untimed: the two learning batches an encoder gets -- the second one a gigabyte after the first -- and their promotions)
            step()
        dt_warm, _ = timed(enc, args.steps)
        enc.set_option(N.OPT_PROMOTE, 0)
        enc.set_option(N.OPT_PROMOTE, 3)
        memo_on[id(enc)] = True
    # the same steps with the piece memo switched off (it neither reads nor fills it): the companion figure `value_no_memo`
    dt_nomemo = None
    nm_steps = args.steps if args.no_memo_steps is None else args.no_memo_steps
    if not args.no_memo and nm_steps > 0:
        # (no memo means no promoted pieces either -- they are memo answers moved into the key tables: dropped for this leg, learnt again after it by two
        #  untimed steps, the two automatic rounds an encoder gets)
        promoted_before = enc.piece_stats().get("promoted_pieces_in_tables", 0)
        enc.set_option(N.OPT_PROMOTE, 0)
        enc.set_option(N.OPT_PROMOTE, 3)
        enc.set_option(N.OPT_PIECE_MEMO, 0)
        memo_on[id(enc)] = False
        step()
        dt_nomemo, _ = timed(enc, nm_steps)
        enc.set_option(N.OPT_PIECE_MEMO, 1)
        enc.set_option(N.OPT_PROMOTE, 1)
        memo_on[id(enc)] = True
        if promoted_before:
            memo_on[id(enc)] = False                              # (kind 6: these two steps must not empty the memo they learn from)
            step(); step()
            memo_on[id(enc)] = True
            fence()
        enc.set_option(N.OPT_PROMOTE, 0)
    # the same corpus under a stand-in vocabulary that has never seen it (tools/train_bpe.py synth100k_heldout: the same size and recipe
    # WITHOUT the bench's generator in the training text): `value_heldout_vocab`.  synth100k is trained on the generator's own output, so
    # its whole-piece hit rate flatters; the real cl100k_base lies somewhere between the two.  Its own encoder, its own memo, the same
    # warm-up; every document against the oracle under that vocabulary as well.
    heldout = None
    ho_steps = args.heldout_steps
    if ho_steps is None:
        ho_steps = args.steps if (world == 1 and args.pattern == 2 and args.vocab is None and not args.no_memo and "VOCAB-SUBSTITUTED" in vocab_name) else 0
    if ho_steps > 0:
        try:
            raw_h, name_h = load_vocab_bytes(args.pattern, "synth100k_heldout")
            enc_h = N.Encoder(N.Vocab(raw_h), args.pattern, device=local_rank)
            ntok_h = prepare(enc_h)
            dt_h, ntok_h = timed(enc_h, ho_steps)
            heldout = {"value": round(total * world * ho_steps / dt_h / 1e6, 1), "unit": "MB/s", "ms_per_step": round(dt_h / ho_steps * 1e3, 3), "vocab": name_h,
                       "vocab_sha256": hashlib.sha256(raw_h).hexdigest(), "tokens_per_gpu": ntok_h, "piece_stats": stats_of(enc_h) if rank == 0 else None,
                       "parity": "unchecked"}
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                from oracle import oracle as O
                tm = {}
                bad, first_bad, otok = O.check_batch(O.Vocab(raw_h), args.pattern, d_bytes[:total].cpu().numpy(), d_offs.cpu().numpy(), d_ids[:ntok_h].cpu().numpy(),
                                                     d_ooffs.cpu().numpy(), threads=max(1, os.cpu_count() or 1), timing=tm)
                heldout["parity"] = ("bit-exact vs oracle on all %d docs (%d tokens)" % (n_docs, otok)) if (bad == 0 and otok == ntok_h) else \
                                    "MISMATCH vs oracle: %d of %d docs differ, first %d" % (bad, n_docs, first_bad)
            del enc_h
            if not emu:
                torch.cuda.empty_cache()
        except Exception as ex:
            heldout = {"error": "%s: %s" % (type(ex).__name__, ex)}
        ntok = step()                          # (the output buffers and the gathered counts are the headline vocabulary's again: the checks below read them)
        fence()
    # ---- `real_text` (default run only: kind 1, N = 1): what the number is worth on REAL, non-repeating text under a REAL table.  The only real
    # vocabulary available offline is gpt2.tiktoken (the reference's model/gpt2.tiktoken); the text is the box's own source and documentation
    # files (real_text_corpus above: every file once, NOT tiled).  gpt2 with its own pattern 1 and with the cl100k pattern (REAL vocabulary x REAL
    # text), and the two cl100k-sized stand-ins on the same text.  Every figure: the memo EMPTIED before every step (a first pass, as a fresh
    # TikTokenizer meets the text), full-batch parity against the oracle, piece statistics.  Companion figures, never `value`.
    real_leg = None
    rt_mb = 256 if args.real_text_mb is None else args.real_text_mb
    if world == 1 and args.kind == 1 and not args.parity_only and rt_mb > 0 and not args.no_memo:
        try:
            r_bytes, r_offs, r_meta = real_text_corpus(rt_mb << 20, 256, 768)
            r_nd, r_total = len(r_offs) - 1, int(r_offs[-1])
            rd_bytes = torch.zeros(r_total + 64, dtype=torch.uint8, device=dev)
            rd_bytes[:r_total] = torch.from_numpy(r_bytes).to(dev)
            rd_offs = torch.from_numpy(r_offs).to(dev)
            rd_ids = torch.empty(r_total, dtype=torch.int32, device=dev)
            rd_ooffs = torch.empty(r_nd + 1, dtype=torch.int64, device=dev)
            r_steps = max(1, min(args.steps, 5))
            real_leg = {"corpus": r_meta, "steps": r_steps, "unit": "MB/s",
                        "piece_memo": "EMPTIED before every timed step and nothing promoted (`value`): real text is not tiled, every step is a first pass; `value_warm_memo`: the "
                                      "memo and the promoted pieces as earlier passes over this very text left them (an upper bound); `value_no_memo`: switched off",
                        "by_vocab": {}}
            combos = [("gpt2", 1), ("gpt2", 2), (None, 2), ("synth100k_heldout", 2)]
            for vname, pat in combos:
                raw_v, label_v = load_vocab_bytes(pat, vname)
                en = N.Encoder(N.Vocab(raw_v), pat, device=local_rank)

                def r_step(mode):
                    if mode == "empty":
                        en.set_option(N.OPT_PIECE_MEMO, 2)
                    return en.encode_batch_device(rd_bytes.data_ptr(), rd_offs.data_ptr(), r_nd, r_total, rd_ids.data_ptr(), r_total, rd_ooffs.data_ptr(), stream)

                def r_timed(mode):
                    device_sync()
                    t0 = time.perf_counter()
                    for _ in range(r_steps):
                        nt = r_step(mode)
                    device_sync()
                    return (time.perf_counter() - t0) / r_steps, nt
                en.set_option(N.OPT_PROMOTE, 0)                     # (a first pass has nothing promoted)
                en.set_option(N.OPT_PIECE_MEMO, 0)
                r_step("off")                                       # (untimed: the workspace takes the size this batch needs)
                t_off, _ = r_timed("off")
                en.set_option(N.OPT_PIECE_MEMO, 1)
                r_step("empty")                                     # (untimed: lists that grow under a miss-heavy table grow here)
                en.set_profiling(True)
                en.kernel_ms(reset=True)
                t_first, r_ntok = r_timed("empty")
                r_kms = en.kernel_ms()
                en.set_profiling(False)
                en.set_option(N.OPT_PROMOTE, 1)                     # warm: the memo kept, and what the encoder promotes out of it (two learning batches, untimed)
                for _ in range(min(6, 2 + int((1 << 30) // max(1, r_total)))):
                    r_step("warm")
                en.adapt_stats()                                    # (waits for the promotion being built behind the last learning batch: the timed steps meet settled tables)
                t_warm, _ = r_timed("warm")
                r_promoted = en.piece_stats().get("promoted_pieces_in_tables", 0)
                en.set_option(N.OPT_PROMOTE, 0)
                en.set_option(N.OPT_PROMOTE, 3)
                en.set_option(N.OPT_PIECE_MEMO, 2)
                en.set_option(N.OPT_PIECE_STATS, 1)
                en.piece_stats(reset=True)
                r_ntok = r_step("empty")
                device_sync()
                r_stats = en.piece_stats(reset=True)
                en.set_option(N.OPT_PIECE_STATS, 0)
                r_stats.pop("batches", None)
                ent = {"vocab": label_v, "pattern": PATTERN_NAME[pat], "value": round(r_total / t_first / 1e6, 1), "ms_per_step": round(t_first * 1e3, 3),
                       "value_warm_memo": round(r_total / t_warm / 1e6, 1), "promoted_pieces_warm": r_promoted, "value_no_memo": round(r_total / t_off / 1e6, 1),
                       "tokens": r_ntok, "bytes_per_token": round(r_total / max(1, r_ntok), 3), "piece_stats": r_stats,
                       "kernels_ms": {k: round(v[0] / max(1, v[1]), 4) for k, v in r_kms.items()}, "parity": "unchecked",
                       "side_by_side_batches": int(en.side_by_side_batches),
                       "kernels_note": "side_by_side_batches > 0: this encoder's batches ran k_merge_long_q and k_merge_coop BESIDE k_merge_short on streams of their own "
                                       "(a previous batch left <= 2^20 long misses): kernels_ms.k_merge_short is then the three side by side, k_merge_long_group what runs in front of them"}
                # the same roofline arithmetic as the headline's, on the first-pass step; counted traffic from profiles/traffic_real_latest.json
                # when it was collected on these kernel sources and this very text
                r_alg = r_total + 4 * int(r_ntok) + 16 * r_nd
                r_traffic, r_tnote = None, "no PMC summary of this text for this build"
                try:
                    tjr = json.load(open(os.path.join(ROOT, "profiles", "traffic_real_latest.json")))
                    if tjr.get("src_sha") == kernel_sources_sha() and tjr.get("corpus_sha256") == r_meta.get("sha256") and tjr.get("vocab_pattern") == "%s/pattern%d" % (vname or "", pat):
                        r_traffic, r_tnote = int(sum(v["hbm_bytes_per_launch"] for v in tjr["by_kernel"].values())), "rocprofv3 PMC passes of this build on this text (profiles/traffic_real_latest.json)"
                except Exception:
                    pass
                ent["roofline"] = {"bound": "hbm", "achieved": round(r_alg / t_first / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(r_alg / t_first / 1e9 / HBM_PEAK_GBPS, 5),
                                   "algorithmic_bytes_per_launch": r_alg, "traffic_pipeline": r_traffic, "wasted": round(r_traffic / r_alg, 3) if r_traffic else None, "traffic_note": r_tnote}
                if not args.no_cpu_baseline:
                    from oracle import oracle as O
                    tm = {}
                    bad, first_bad, otok = O.check_batch(O.Vocab(raw_v), pat, r_bytes, r_offs, rd_ids[:r_ntok].cpu().numpy(), rd_ooffs.cpu().numpy(),
                                                         threads=max(1, os.cpu_count() or 1), timing=tm)
                    ent["parity"] = ("bit-exact vs oracle on all %d docs (%d tokens)" % (r_nd, otok)) if (bad == 0 and otok == r_ntok) else \
                                    "MISMATCH vs oracle: %d of %d docs differ, first %d" % (bad, r_nd, first_bad)
                    ent["cpu_oracle_all_threads_mbps"] = round(r_total / tm["seconds"] / 1e6, 1)
                real_leg["by_vocab"]["%s/pattern%d" % (vname or VOCAB_OF_PATTERN[pat][0], pat)] = ent
                # ---- DRIFT (round 6, TKZ_OPT_ADAPT): ONE encoder whose text changes under it.  An encoder that has learnt the headline's synthetic text
                #      meets the real text (and the reverse): GB/s of every step after the change, against an encoder that only ever saw the second text,
                #      both on repeated passes with memo and promotions left as they come (`value_warm_memo`'s conditions).  gpt2 / pattern 1 only.
                if vname == "gpt2" and pat == 1 and not args.no_cpu_baseline and seed is not None:
                    try:
                        def rate_steps(e2, fn, nbytes, nsteps):
                            out = []
                            for _ in range(nsteps):
                                device_sync(); t0 = time.perf_counter(); fn(e2); device_sync()
                                out.append(round(nbytes / (time.perf_counter() - t0) / 1e6, 1))
                            return out
                        sl_docs = min(n_docs, 2_000_000)                                   # a ~1 GB slice of the headline corpus: a step of the synthetic side
                        sl_total = int(d_offs[sl_docs].item())

                        def syn(e2):
                            return e2.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), sl_docs, sl_total, d_ids.data_ptr(), sl_total, d_ooffs.data_ptr(), stream)

                        def real(e2):
                            return e2.encode_batch_device(rd_bytes.data_ptr(), rd_offs.data_ptr(), r_nd, r_total, rd_ids.data_ptr(), r_total, rd_ooffs.data_ptr(), stream)
                        drift = {"unit": "MB/s per step", "synthetic_step_bytes": sl_total, "real_step_bytes": r_total}
                        for name, first, first_b, second, second_b in (("synthetic_then_real", syn, sl_total, real, r_total), ("real_then_synthetic", real, r_total, syn, sl_total)):
                            e_d = N.Encoder(N.Vocab(raw_v), pat, device=local_rank)        # default options: promotions automatic, TKZ_OPT_ADAPT on
                            for _ in range(max(3, int((3 << 30) // max(1, first_b)))):
                                first(e_d)                                                 # ~3 GB of the first text: learnt, promoted ...
                            # ... and SETTLED: a promotion is built on a host thread (tens of milliseconds -- these batches take 3..7 ms each), so the text goes on
                            # until what was learnt from it is in the tables and the miss share has a level (an encoder that has LEARNT the first text meets the second:
                            # a change that falls into the build itself is the case tools/adapt_probe.py's `fast` mode shows)
                            extra_first = 0
                            while e_d.adapt_stats()["settled_miss_share"] is None and extra_first < 40:
                                first(e_d); device_sync(); time.sleep(0.005); extra_first += 1
                            before = e_d.adapt_stats()
                            n2 = max(7, int((3 << 30) // max(1, second_b)))
                            series = rate_steps(e_d, second, second_b, n2)
                            after = e_d.adapt_stats()
                            e_f = N.Encoder(N.Vocab(raw_v), pat, device=local_rank)        # ... and one that only ever sees the second text
                            fresh = rate_steps(e_f, second, second_b, n2)
                            k2gb = min(n2, max(1, int((2 << 30) // max(1, second_b))))     # steps within 2 GB of the change
                            # (the MEDIAN of the steps beyond 2 GB: a step that happens to be a learning batch, or to run beside a promotion being built, is slower on
                            #  either encoder and falls on different steps of the two)
                            tail_d, tail_f = sorted(series[k2gb:] or series[-1:]), sorted(fresh[k2gb:] or fresh[-1:])
                            med_d, med_f = tail_d[len(tail_d) // 2], tail_f[len(tail_f) // 2]
                            drift[name] = {"mbps_by_step_after_the_change": series, "mbps_by_step_fresh_encoder": fresh,
                                           "value_after_drift": med_d, "value_fresh": med_f,
                                           "ratio": round(med_d / med_f, 3), "steps_within_2GB": k2gb, "first_text_batches_until_settled": extra_first,
                                           "relearns": after["relearns"] - before["relearns"], "promoted_before": before["promoted_pieces"], "promoted_after": after["promoted_pieces"],
                                           "miss_share_settled_before": before["settled_miss_share"], "miss_share_recent_after": after["recent_miss_share"]}
                            del e_d, e_f
                        real_leg["drift"] = drift
                    except Exception as ex:
                        real_leg["drift"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                del en
            del rd_bytes, rd_offs, rd_ids, rd_ooffs, r_bytes, r_offs
            if not emu:
                torch.cuda.empty_cache()
        except Exception as ex:                                      # an auxiliary figure must never cost the bench line
            real_leg = {"error": "%s: %s" % (type(ex).__name__, ex)}
        ntok = step()                                                # (the output buffers hold the headline batch's result again)
        fence()
    # ---- `first_call`: what a job that encodes ONE batch pays.  A FRESH encoder, its workspace reserved at construction (tkz_encoder_reserve: the
    # reference pays construction costs in CreateTokenizer, TokenizerBuilder.cs:210-213), then the bench batch ONCE, timed, with nothing untimed before
    # it: empty memo, nothing promoted, cold tables, the sizing attempt and the learning window's counting inside the call.  `value` never includes it.
    first_call = None
    if world == 1 and not args.parity_only and not args.no_memo and not args.no_first_call:
        try:
            e_fc = N.Encoder(vocab, args.pattern, device=local_rank)
            device_sync(); t0 = time.perf_counter()
            e_fc.reserve(total, n_docs)
            device_sync(); t_res = time.perf_counter() - t0
            t0 = time.perf_counter()
            nt_fc = e_fc.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, total, d_ids.data_ptr(), total, d_ooffs.data_ptr(), stream)
            device_sync(); t_first_call = time.perf_counter() - t0
            t0 = time.perf_counter()
            e_fc.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, total, d_ids.data_ptr(), total, d_ooffs.data_ptr(), stream)
            device_sync(); t_second = time.perf_counter() - t0
            first_call = {"first_call_ms": round(t_first_call * 1e3, 3), "second_call_ms": round(t_second * 1e3, 3), "reserve_ms": round(t_res * 1e3, 1),
                          "workspace_bytes": int(e_fc.workspace_bytes), "tokens": int(nt_fc),
                          "note": "a fresh encoder: tkz_encoder_reserve(batch size) at construction (reserve_ms: its hipMallocs), then the bench batch once -- empty memo, nothing "
                                  "promoted, sizing attempt and learning window inside the call -- and once more"}
            del e_fc
            if not emu:
                torch.cuda.empty_cache()
        except Exception as ex:
            first_call = {"error": "%s: %s" % (type(ex).__name__, ex)}
        ntok = step()
        fence()
    if warm is not None:
        del warm, w_bytes, w_offs, w_ids
        if not emu:
            torch.cuda.empty_cache()
    # the same steps two at a time through tkz_encode_batch_device_begin / _end (two streams, two output buffers, two workspaces of the
    # encoder): what keeping batches in flight buys over one synchronous call after the other -- a companion figure, never `value`
    dt_pipe = None
    pipe_note = None
    if args.pipelined_steps > 0 and args.kind != 5 and not emu:
        # every batch in flight has its own {docs, bytes, tokens} block (tkz_encode_batch_device_begin_counts) and its own gathered table: the
        # count all-gather of a batch is enqueued on that batch's stream behind its _end, at any N.  The untimed pass (the second workspace
        # takes its size) runs WITHOUT the collective, and the ranks agree that every one of them got through it before any enters a
        # gather: a rank that could not allocate must not leave the others waiting inside RCCL.
        pair = None
        ok_local = 0
        try:
            s2 = torch.cuda.Stream()
            d_ids2 = torch.empty_like(d_ids); d_ooffs2 = torch.empty_like(d_ooffs)
            d_cnt = torch.zeros(2, 3, dtype=torch.int64, device=dev)
            d_tab = torch.zeros(2, world * 3, dtype=torch.int64, device=dev)
            outs = [(d_ids, d_ooffs, stream, d_cnt[0], d_tab[0]), (d_ids2, d_ooffs2, s2.cuda_stream, d_cnt[1], d_tab[1])]

            def pair(gather=True):
                hs = [enc.encode_batch_device_begin(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, total, o[0].data_ptr(), total, o[1].data_ptr(), o[2],
                                                    d_counts3=o[3].data_ptr()) for o in outs]
                res = []
                for h, o in zip(hs, outs):
                    res.append(enc.encode_batch_device_end(h))
                    if comm is not None and gather:
                        comm.gather_async(o[3].data_ptr(), o[2], d_table=o[4].data_ptr())
                return res
            ok_local = 1 if pair(gather=False) == [ntok, ntok] else 0
            if not ok_local:
                pipe_note = "the untimed pass gave another token count"
        except Exception as ex:                                # (e.g. no room for a second workspace)
            ok_local = 0
            pipe_note = "%s: %s" % (type(ex).__name__, ex)
        if world > 1:
            t = torch.tensor([ok_local], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok_local = int(t.item())
        if ok_local:
            fence()
            t0 = time.perf_counter()
            for _ in range((args.pipelined_steps + 1) // 2):
                pair()
            fence()
            dt_pipe = (time.perf_counter() - t0) / (2 * ((args.pipelined_steps + 1) // 2))
            if world > 1:
                t = torch.tensor([dt_pipe], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_pipe = float(t.item())
            assert torch.equal(d_ids2[:ntok], d_ids[:ntok]) and torch.equal(d_ooffs2, d_ooffs)
            assert d_cnt.cpu().tolist() == [[n_docs, total, ntok]] * 2
            if comm is not None:
                assert d_tab[0].cpu().tolist()[3 * rank:3 * rank + 3] == [n_docs, total, ntok] and torch.equal(d_tab[0], d_tab[1])
            del d_ids2, d_ooffs2
        elif pipe_note is None:
            pipe_note = "skipped: the untimed pass failed on another rank"
    g = comm.result() if comm is not None else sharded.gather_counts(n_docs, total, ntok)
    n_tokens_rank = int(g["table"][rank][2])
    assert n_tokens_rank == ntok and int(g["table"][rank][0]) == n_docs and int(g["table"][rank][1]) == total

    shard_note = None
    if args.write_shards:
        os.makedirs(args.write_shards, exist_ok=True)
        path = os.path.join(args.write_shards, "tokens.%05d.tkzs" % rank)
        ts = time.perf_counter()
        N.shard_write_device(path, d_ids.data_ptr(), ntok, d_ooffs.data_ptr(), n_docs, g["doc_base"], g["token_base"], device=local_rank)
        shard_note = {"file": path, "bytes": os.path.getsize(path), "seconds": round(time.perf_counter() - ts, 3)}

    # ---- N > 1: parity on every rank.  Each rank checks the first and the last --parity-sample-docs documents of ITS OWN shard against the oracle
    # (tkzo_check_batch on its share of the host's CPUs, all ranks at the same time), the verdicts are reduced, and rank 0's line says
    # "bit-exact on N x ... sampled docs" or names the first bad (rank, document).  Nothing of this is inside a timed region.
    multi = None
    if world > 1 and not args.no_cpu_baseline:
        verdict = [0, 0, 0, 0, 1 << 62, 0.0]                     # bad docs, docs checked, tokens, ranks that could not check, first bad (rank << 40 | doc), seconds
        note_err = None
        try:
            from oracle import oracle as O
            ncpu = max(1, os.cpu_count() or 1)
            try:
                q = open("/sys/fs/cgroup/cpu.max").read().split()
                if q[0] != "max":
                    ncpu = max(1, min(ncpu, int(float(q[0]) / float(q[1]) + 0.5)))
            except Exception:
                pass
            th = max(1, ncpu // world)
            res = check_ranges(O, O.Vocab(raw), args.pattern, d_bytes, d_offs, d_ids, d_ooffs, ntok, sample_ranges(n_docs, args.parity_sample_docs), th)
            same_counts = int(d_ooffs[0].item()) == 0 and int(d_ooffs[n_docs].item()) == ntok
            if not same_counts and res["bad"] == 0:
                res["bad"], res["first_bad"] = 1, 0
            verdict = [res["bad"], res["docs"], res["tokens"], 0, ((rank << 40) | max(0, res["first_bad"])) if res["bad"] else (1 << 62), res["seconds"]]
        except Exception as ex:                                  # (e.g. no C compiler for the oracle on this host: the line says so instead of claiming parity)
            verdict[3] = 1
            note_err = "%s: %s" % (type(ex).__name__, ex)
        tsum = torch.tensor(verdict[:4], dtype=torch.int64, device=dev)
        tmin = torch.tensor([verdict[4]], dtype=torch.int64, device=dev)
        tmax = torch.tensor([verdict[5]], dtype=torch.float64, device=dev)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        bad_all, docs_all, tok_all, failed = (int(x) for x in tsum.cpu().tolist())
        first_code = int(tmin.item())
        per = min(args.parity_sample_docs, n_docs) if n_docs > 2 * args.parity_sample_docs else None
        where = ("the first and the last %d documents of every rank's shard" % per) if per else "every document of every rank's shard"
        if failed:
            note = "unchecked on %d of %d ranks (CPU oracle unavailable%s); %d sampled docs checked elsewhere, %d differ" % (failed, world, ": " + note_err if note_err else "", docs_all, bad_all)
        elif bad_all:
            note = "MISMATCH vs oracle: %d of %d sampled docs differ; first: rank %d, document %d of its shard" % (bad_all, docs_all, first_code >> 40, first_code & ((1 << 40) - 1))
        else:
            note = ("bit-exact vs oracle on %d x %d = %d sampled docs (%s; %d tokens; every rank's offsets start at 0 and end at its token count; "
                    "all ranks at once, %.1f s for the slowest)" % (world, docs_all // world, docs_all, where, tok_all, float(tmax.item())))
        multi = {"note": note, "bad": bad_all, "docs": docs_all, "tokens": tok_all, "ranks_unchecked": failed}
        # the other ranks sleep on the rendezvous store (a blocking socket wait, no spinning) while rank 0 times the CPU baseline on the host's cores
        store = dist.distributed_c10d._get_default_store()
        if rank != 0:
            try:
                import datetime
                store.wait(["tkz_bench_cpu_baseline_done"], datetime.timedelta(seconds=1200))
            except Exception:                                    # (rank 0 died before its line: nothing to wait for)
                pass

    if rank == 0:
        job_bytes, job_tokens, job_docs = g["bytes"], g["tokens"], g["docs"]
        ms_per_step = dt / args.steps * 1e3
        value = job_bytes * args.steps / dt / 1e6
        # ---- roofline of the dominant kernel (HBM-bound integer/indexing work; no MFMA) ----
        dom = max(kms, key=lambda k: kms[k][0])
        dom_ms = kms[dom][0] / max(1, kms[dom][1])
        # read text + write int32 ids + 8 B offset in + 8 B offset out -- of the rank whose kernels are priced (the slowest one at N > 1)
        alg_bytes = int(g["table"][kms_rank][1]) + 4 * int(g["table"][kms_rank][2]) + 16 * int(g["table"][kms_rank][0])
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_pipeline, traffic_by_kernel, traffic_note = None, None, None, "no PMC summary for this build"
        issue = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("src_sha") != kernel_sources_sha():
                    traffic_note = "profiles/traffic_latest.json was collected on other kernel sources (src_sha differs): not reported"
                elif tj.get("docs_per_gpu") != n_docs or tj.get("kind") != args.kind or dom not in tj.get("by_kernel", {}):
                    traffic_note = "profiles/traffic_latest.json is for another workload / kernel: not reported"
                else:
                    traffic, traffic_note = tj["by_kernel"][dom]["hbm_bytes_per_launch"], "rocprofv3 PMC passes of this build (profiles/traffic_latest.json)"
                    traffic_by_kernel = {k: v["hbm_bytes_per_launch"] for k, v in tj["by_kernel"].items()}
                    traffic_pipeline = int(sum(traffic_by_kernel.values()))
                    # the binding limit in the record, not only in prose: the dominant kernel's VALU wave-instructions of the same PMC passes against the chip's
                    # issue rate (256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz = 0.614 T/s), and the lanes they had switched on
                    iss = tj["by_kernel"][dom].get("issue")
                    if iss and iss.get("valu_insts"):
                        rate = iss["valu_insts"] / (dom_ms * 1e-3) / 1e12
                        issue = {"kernel": dom, "valu_per_kib": round(iss["valu_insts"] / (int(g["table"][kms_rank][1]) / 1024.0), 1), "valu_rate_tps": round(rate, 3),
                                 "frac_of_valu_peak": round(rate / VALU_PEAK_TPS, 3), "lanes_active": iss.get("lanes_active"), "peak_tps": VALU_PEAK_TPS,
                                 "salu_per_kib": round(iss["salu_insts"] / (int(g["table"][kms_rank][1]) / 1024.0), 1) if iss.get("salu_insts") else None,
                                 "by_kernel": {k: v.get("issue") for k, v in tj["by_kernel"].items() if v.get("issue")},
                                 "note": "SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_THREAD_CYCLES_VALU of the rocprofv3 PMC passes of this build (profiles/traffic_latest.json), "
                                         "per launch; rate = instructions / this run's launch duration"}
            except Exception:
                traffic = None
        leftovers = enc.pretok_leftovers() if hasattr(enc, "pretok_leftovers") else (0, 0)
        pipe_achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9      # (ms_per_step: the slowest rank's; alg_bytes: that rank's)
        # `achieved` / `frac` price the WHOLE step (every kernel of the launch sequence has to move its share of the algorithmic bytes: crediting
        # the dominant kernel alone with all of them flatters); the dominant kernel's own figure is beside it (`*_dominant`).  `traffic` is the
        # dominant kernel's counted HBM bytes per launch, `traffic_pipeline` all kernels' together, `wasted` = traffic_pipeline / algorithmic.
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(pipe_achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(pipe_achieved / HBM_PEAK_GBPS, 5),
                    "achieved_dominant": round(achieved, 2), "frac_dominant": round(achieved / HBM_PEAK_GBPS, 5),
                    "traffic": traffic, "traffic_pipeline": traffic_pipeline,
                    "wasted": round(traffic_pipeline / alg_bytes, 3) if traffic_pipeline else None, "traffic_by_kernel": traffic_by_kernel, "traffic_note": traffic_note,
                    "issue": issue,
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_ms, 4), "rank": kms_rank,
                    "kernels_ms": {k: round(v[0] / max(1, v[1]), 4) for k, v in kms.items()},
                    "side_by_side_batches": int(enc.side_by_side_batches) if hasattr(enc, "side_by_side_batches") else None,
                    "note": "achieved = algorithmic bytes (SURVEY.md 8d: text + 4 B per id + 16 B per document) / the step's time; *_dominant = the same bytes / "
                            "the dominant kernel's average launch duration (HIP events on the launch stream)"}
        if args.pattern in (3, 4):     # of the 4 KiB blocks: handed on by the ASCII block scanner / by the multi-byte one as well (to the sequential matcher)
            roofline["o200k_blocks"] = {"total": (total + 3967) // 3968, "after_ascii_scanner": leftovers[0], "after_multibyte_scanner": leftovers[1]}
        # ---- CPU baseline (the oracle = reference-algorithm restatement, "port") + parity on the sample ----
        cpu = None
        host_path = None
        parity_note = "unchecked"
        host_api = None
        if multi is not None:
            # N > 1: every rank has checked the first and the last --parity-sample-docs documents of ITS OWN shard (above, all ranks at once); the CPU
            # baseline is rank 0's alone, on a bounded sample of its shard, while the other ranks sleep on the store
            parity_note = multi["note"]
        if not args.no_cpu_baseline and (world == 1 or multi is not None):
            try:
                from oracle import oracle as O
                ov = O.Vocab(raw)
                ncpu = max(1, os.cpu_count() or 1)
                if world == 1:
                    # ---- parity: EVERY document of the batch against the oracle, on all host cores (tkzo_check_batch encodes each
                    # document and compares it in place with the ids the GPU left for it: nothing beyond the inputs is allocated) ----
                    h_offs = d_offs.cpu().numpy()
                    h_bytes = d_bytes[:total].cpu().numpy()
                    h_ooffs = d_ooffs.cpu().numpy()
                    h_ids = d_ids[:n_tokens_rank].cpu().numpy()
                    same = int(h_ooffs[0]) == 0 and int(h_ooffs[n_docs]) == n_tokens_rank and bool((np.diff(h_ooffs) >= 0).all())
                    tm = {}
                    if n_docs >= 4 * ncpu:
                        bad, first_bad, otok = O.check_batch(ov, args.pattern, h_bytes, h_offs, h_ids, h_ooffs, threads=ncpu, timing=tm)
                        best = (total / tm["seconds"] / 1e6, ncpu, total, n_docs)
                    else:                                         # (one giant document: a single thread, no sweep)
                        bad, first_bad, otok = O.check_batch(ov, args.pattern, h_bytes, h_offs, h_ids, h_ooffs, threads=1, timing=tm)
                        best = (total / tm["seconds"] / 1e6, 1, total, n_docs)
                    same = same and bad == 0 and otok == n_tokens_rank
                    parity_note = ("bit-exact vs oracle on all %d docs (%d tokens, %.1f s on %d host threads); offsets monotone, ending at the token count"
                                   % (n_docs, otok, tm["seconds"], best[1])) if same else "MISMATCH vs oracle: %d of %d docs differ, first %d" % (bad, n_docs, first_bad)
                    if args.parity_only:
                        raise StopIteration
                    covers = "the %d-thread run covers the whole batch and is the parity check" % ncpu
                else:
                    # (rank 0's sample: the first --cpu-sample-docs documents of its shard; the all-thread run of the sweep is its first point)
                    ns0 = min(args.cpu_sample_docs, n_docs)
                    h_offs = d_offs[:ns0 + 1].cpu().numpy()
                    h_bytes = d_bytes[:int(h_offs[-1])].cpu().numpy()
                    h_ooffs = d_ooffs[:ns0 + 1].cpu().numpy()
                    h_ids = d_ids[:int(h_ooffs[-1])].cpu().numpy()
                    tm = {}
                    bad0, _, _ = O.check_batch(ov, args.pattern, h_bytes, h_offs, h_ids, h_ooffs, threads=ncpu, timing=tm)
                    best = (len(h_bytes) / tm["seconds"] / 1e6, ncpu, len(h_bytes), ns0)
                    if bad0:
                        parity_note += "; MISMATCH in rank 0's CPU-baseline sample: %d documents" % bad0
                    covers = "every run on rank 0's first %d documents, the other ranks asleep" % ns0
                # ---- CPU baseline: the same restatement timed on the host cores.  The all-core run above is the first point; fewer
                # threads (SMT siblings and memory channels decide which count is best) on a bounded sample; and one thread ----
                ns = min(args.cpu_sample_docs, n_docs)
                nb = int(h_offs[ns])
                sweep = {str(best[1]): round(best[0], 1)}
                for th in sorted({max(1, ncpu // 2), max(1, ncpu // 4)} - {best[1]}):
                    if n_docs < 4 * ncpu:
                        break
                    O.check_batch(ov, args.pattern, h_bytes[:nb], h_offs[:ns + 1], h_ids, h_ooffs[:ns + 1], threads=th, timing=tm)
                    sweep[str(th)] = round(nb / tm["seconds"] / 1e6, 1)
                    if nb / tm["seconds"] / 1e6 > best[0]:
                        best = (nb / tm["seconds"] / 1e6, th, nb, ns)
                n1 = max(1, min(ns // 10, 200_000)) if n_docs > 1 else 1
                nb1 = int(h_offs[n1]) if n_docs > 1 else min(total, 32 << 20)
                if n_docs > 1:
                    O.check_batch(ov, args.pattern, h_bytes[:nb1], h_offs[:n1 + 1], h_ids, h_ooffs[:n1 + 1], threads=1, timing=tm)
                    cpu_1t = round(nb1 / tm["seconds"] / 1e6, 2)
                else:
                    cpu_1t = round(best[0], 2)
                quota, eff = None, None
                try:
                    quota = open("/sys/fs/cgroup/cpu.max").read().strip()
                    q = quota.split()
                    if q[0] != "max":
                        eff = float(q[0]) / float(q[1])
                except Exception:
                    pass
                # `cores` = the CPUs the run could really use: the cgroup's quota when there is one (the box shows 256 hardware threads and grants 16
                # CPUs of time), else the thread count; `threads` = how many threads the best run used
                cores_eff = min(best[1], eff) if eff else best[1]
                cpu = {"value": round(best[0], 2), "unit": "MB/s", "cores": int(cores_eff) if float(cores_eff).is_integer() else round(cores_eff, 2), "threads": best[1],
                       "kind": "port", "value_1_thread": cpu_1t, "by_threads": sweep,
                       "host_threads_available": ncpu, "cgroup_cpu_max": quota,
                       "host_parallel_speedup": O.host_parallelism(sorted({1, max(1, ncpu // 4), max(1, ncpu // 2), ncpu})),
                       "sample": "%d documents (%.1f MB) of the same corpus on %d host threads (`cores`: the CPUs the cgroup grants them) -- the best of the thread counts tried (by_threads: MB/s; "
                                 "%s); one thread: the first %d documents; reference-algorithm CPU restatement "
                                 "(oracle/), 8192-entry LRU memo and reusable scratch per thread" % (best[3], best[2] / 1e6, best[1], covers, n1)}
                if world > 1 or emu:
                    raise StopIteration              # (the PCIe-inclusive and host-API legs are single-GPU companions)
                ns = min(ns, n_docs)
                # the real C# TokenizerLib beside it, when this host has a .NET SDK and a reference checkout (never in this image)
                if shutil.which("dotnet") and os.environ.get("TKZ_REFERENCE_DIR"):
                    sp = "/tmp/tkz_bench_sample.bin"
                    nd = min(ns, 200_000)
                    with open(sp, "wb") as f:
                        f.write(np.int64(nd).tobytes()); f.write(h_offs[:nd + 1].tobytes()); f.write(h_bytes[:int(h_offs[nd])].tobytes())
                    open("/tmp/tkz_bench_vocab.tiktoken", "wb").write(raw)
                    from tokenizer_amd import tokenizer as TK
                    os.environ["TKZ_BENCH_VOCAB"] = "/tmp/tkz_bench_vocab.tiktoken"
                    os.environ["TKZ_BENCH_PATTERN"] = {1: TK.REGEX_PATTERN_1, 2: TK.REGEX_CL100K, 3: TK.REGEX_O200K, 4: TK.REGEX_O200K}[args.pattern]
                    cpu["reference_dotnet"] = reference_dotnet_baseline(sp)
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
                    host_same = True
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
                        best_rate = 0.0                                                   # (the better of two runs: one 12 ms call is easily disturbed)
                        for _rep in range(2):
                            tc = time.perf_counter()
                            r_ids, r_ooffs = enc.encode_batch(bufs[0], bufs[1], out=bufs[2])
                            best_rate = max(best_rate, len(hh_bytes) / (time.perf_counter() - tc) / 1e6)
                        rates.append(round(best_rate, 1))
                        host_same = host_same and int(r_ooffs[-1]) == int(d_ooffs[nh].item()) and np.array_equal(r_ids[:len(h_ids)], h_ids[:len(r_ids)])
                    host_path = {"value": rates[0], "value_pinned_buffers": rates[1], "unit": "MB/s", "docs": nh, "same_ids_as_device_path": bool(host_same),
                                 "engine_downloads": int(enc.engine_downloads) if hasattr(enc, "engine_downloads") else None,
                                 "note": "tkz_encode_batch_utf8 on host buffers: H2D of the text, kernels and D2H of ids and offsets; chunks of 16 MB (page-locked) / 32 MB, two launch sequences enqueued ahead, page-locked results downloaded by a copy engine of their own (engine_downloads: how many such copies this encoder has made); the better of two calls"}
                except Exception as ex:                      # an auxiliary figure must never cost the bench line
                    host_path = {"error": "%s: %s" % (type(ex).__name__, ex)}
                # The ITokenizer-shaped surface: tkz::TikTokenizer::EncodeBatchFlat(std::vector<std::string>) of include/tkz_tokenizer.hpp on the
                # same documents held as strings -- a gather into page-locked memory by host threads, then the host-buffer entry point.  A C++
                # program (tests/cpp/bench_host_api.cpp) built here with g++ against the same libtkz.so; its ids are compared by checksum.
                try:
                    import tempfile
                    from tokenizer_amd import tokenizer as TK
                    tdir = tempfile.mkdtemp(prefix="tkz_host_api_")
                    exe = os.path.join(tdir, "bench_host_api")
                    libdir = os.path.dirname(enc.lib.path)
                    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "bench_host_api.cpp"),
                                           "-L", libdir, "-ltkz", "-Wl,-rpath," + libdir, "-o", exe])
                    open(os.path.join(tdir, "v.tiktoken"), "wb").write(raw)
                    open(os.path.join(tdir, "regex.txt"), "w").write({1: TK.REGEX_PATTERN_1, 2: TK.REGEX_CL100K, 3: TK.REGEX_O200K, 4: TK.REGEX_O200K}[args.pattern])
                    with open(os.path.join(tdir, "sample.bin"), "wb") as f:
                        f.write(np.int64(nh).tobytes()); f.write(hh_offs.astype(np.int64).tobytes()); f.write(hh_bytes.tobytes())
                    out = subprocess.run([exe, os.path.join(tdir, "v.tiktoken"), os.path.join(tdir, "regex.txt"), os.path.join(tdir, "sample.bin"), "0"],
                                         capture_output=True, text=True, timeout=600)
                    shutil.rmtree(tdir, ignore_errors=True)
                    if out.returncode != 0 or args.pattern == 3:       # (pattern 3 is not what the C++ mirror makes of the o200k string: it reads it as .NET does)
                        host_api = {"error": (out.stderr or out.stdout)[-300:] if out.returncode else "the C++ mirror reads the o200k string with the C# engine's semantics (pattern 4)"}
                    else:
                        host_api = json.loads(out.stdout.strip().splitlines()[-1])
                        want = h_ids[:int(h_ooffs[nh])].astype(np.uint32).astype(np.uint64) + np.uint64(1)      # (the driver's position-weighted sum mod 2^64)
                        wsum = int((want * (np.arange(len(want), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1))).sum(dtype=np.uint64))
                        host_api["same_ids_as_device_path"] = bool(host_api["tokens"] == len(want) and host_api["ids_checksum"] == "%016x" % wsum)
                        if isinstance(host_api.get("utf16"), dict):
                            u = host_api["utf16"]
                            u["same_ids_as_device_path"] = bool(u.get("tokens") == len(want) and u.get("ids_checksum") == "%016x" % wsum)
                            u["unit"] = "MB/s of UTF-8 text (the input is twice that in UTF-16 code units)"
                            u["note"] = ("tkz::TikTokenizer::EncodeBatchFlatUtf16 on the same documents as std::u16strings: threaded gather of the code units into page-locked memory + ONE "
                                         "tkz_encode_batch_utf16 (chunked upload, Encoding.UTF8.GetBytes on the device); value_as_csharp: the call sequence of "
                                         "bindings/csharp/GpuTikTokenizer.EncodeBatchFlatPinned replayed in C++ (tests/cpp/bench_host_api.cpp) -- sub-batches of 64 M code units, two "
                                         "page-locked unit buffers, the gather of sub-batch k + 1 on all cores beside the device call of sub-batch k, every sub-batch's ids written "
                                         "straight into ONE page-locked id buffer sized by the densest batch seen; value_as_csharp_managed_array: EncodeBatchFlat, i.e. plus a fresh "
                                         "zero-filled int array of exactly the ids' number filled by all cores; the C# file itself cannot be compiled here")
                        host_api["note"] = ("tkz::TikTokenizer::EncodeBatchFlat on %d std::strings (include/tkz_tokenizer.hpp): threaded gather into page-locked memory + "
                                            "tkz_encode_batch_utf8 + ids left in page-locked memory; the best of %d calls" % (nh, host_api.get("reps", 0)))
                except Exception as ex:
                    host_api = {"error": "%s: %s" % (type(ex).__name__, ex)}
            except StopIteration:
                pass
            except Exception as ex:                          # (e.g. no C compiler for the oracle on this host)
                parity_note = "unchecked: CPU oracle unavailable (%s: %s)" % (type(ex).__name__, ex)
        workloads = {1: "BASELINE.json configs[1]: cl100k_base pattern, %d synthetic ASCII docs/GPU, %d..%d B (mean %.0f), device-resident",
                     2: "BASELINE.json configs[2] shape: mixed UTF-8 (CJK + emoji) corpus, %d docs/GPU, %d..%d B (mean %.0f), device-resident",
                     3: "BASELINE.json configs[4] shape: long-context docs with long single-class runs, %d docs/GPU, %d..%d B (mean %.0f), device-resident",
                     4: "real source text: the reference's lib.rs.txt tiled, %d docs/GPU, %d..%d B (mean %.0f), device-resident",
                     5: "the reference's own benchmark shape (PerfBenchmark/Program.cs:14-32): %d document/GPU of words of the 4096-word table joined by single spaces, "
                        "one Encode call, %d..%d B (mean %.0f), device-resident",
                     6: "REAL text: the source / documentation files of the box (every file once, sorted, NOT tiled; sha256 in config.real_text), %d docs/GPU, "
                        "%d..%d B (mean %.0f), device-resident"}
        if args.kind == 1 and world > 1:
            workloads[1] = ("BASELINE.json configs[3]: cl100k_base pattern, %d synthetic ASCII docs/GPU (" + str(n_docs * world) + " documents sharded over " + str(world) +
                            " GPUs), %d..%d B (mean %.0f), device-resident")
        line = {
            "metric": "input MB/s encoded (cl100k_base)", "value": round(value, 1), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": ("EMULATED: the kernels ran on the CPU emulator of tests/hostemu (a test of this script, not a measurement)" if emu else
                     "real text of the box (not tiled)" if args.kind == 6 else "synthetic"),
            "config": {"workload": workloads[args.kind] % (n_docs, args.min_len, args.max_len, job_bytes / max(1, job_docs)),
                       "pattern": PATTERN_NAME[args.pattern], "vocab_pattern_key": "%s/pattern%d" % (args.vocab or VOCAB_OF_PATTERN[args.pattern][0], args.pattern),
                       "piece_memo": memo_note,
                       "promoted_pieces": promoted_note,
                       "vocab": vocab_name, "vocab_keys": len(vocab), "vocab_sha256": hashlib.sha256(raw).hexdigest(), "docs_per_gpu": n_docs, "bytes_per_gpu": total, "tokens_per_gpu": n_tokens_rank,
                       "job_docs": job_docs, "job_bytes": job_bytes, "job_tokens": job_tokens,
                       "partitioning": "contiguous document ranges, one process per GPU; one all-gather of 3 int64 counts per rank per step"},
            "comm": comm_info,
            "tokens_per_s": round(job_tokens * args.steps / dt, 1),
            "piece_stats": piece_stats,
            "value_no_memo": round(job_bytes * nm_steps / dt_nomemo / 1e6, 1) if dt_nomemo else None,
            "value_warm_memo": round(job_bytes / dt_warm * args.steps / 1e6, 1) if dt_warm else None,
            "value_real_text": (real_leg.get("by_vocab", {}).get("gpt2/pattern1", {}).get("value") if real_leg else None),
            "first_call_ms": first_call.get("first_call_ms") if first_call else None,
            "first_call": first_call,
            "value_after_drift": ((real_leg.get("drift") or {}).get("synthetic_then_real", {}).get("value_after_drift") if real_leg else None),
            "real_text_roofline": (real_leg.get("by_vocab", {}).get("gpt2/pattern1", {}).get("roofline") if real_leg else None),
            "real_text": real_leg,
            "value_heldout_vocab": heldout["value"] if heldout and "value" in heldout else None,
            "heldout_vocab": heldout,
            "value_two_in_flight": round(job_bytes / dt_pipe / 1e6, 1) if dt_pipe else None,
            "two_in_flight_note": pipe_note,
            "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
            "parity": parity_note,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "pcie_inclusive": host_path,
            "value_host_api": host_api["value"] if host_api and "value" in host_api else None,
            "value_host_api_utf16": (host_api.get("utf16") or {}).get("value") if host_api else None,
            "value_host_api_utf16_as_csharp": (host_api.get("utf16") or {}).get("value_as_csharp") if host_api else None,
            "value_host_api_utf16_as_csharp_managed_array": (host_api.get("utf16") or {}).get("value_as_csharp_managed_array") if host_api else None,
            "host_api": host_api,
        }
        if real_meta:
            line["config"]["real_text"] = real_meta
        if emu:
            line["emulated"] = True
        if shard_note:
            line["shard_file"] = shard_note
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        if multi is not None:
            dist.distributed_c10d._get_default_store().set("tkz_bench_cpu_baseline_done", "1")
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
