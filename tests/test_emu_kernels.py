"""`not gpu` coverage of the DEVICE code: the real kernel sources (tokenizer_amd/csrc/tkz_kernels.hip and
the C ABI above them) compiled against the fiber SIMT emulator of tests/hostemu/ and compared with the
oracle.  Sizes are small (the emulator runs one workgroup at a time); the same checks run at full size
on the GPU in test_gpu_parity.py."""
import gzip
import json
import os

import numpy as np
import pytest

import emu
import parity
from conftest import load_golden_json
from tokenizer_amd import _native as N


@pytest.fixture(scope="module")
def lib():
    return emu.library()


@pytest.fixture(scope="module")
def vocab(lib, gpt2_tiktoken_bytes):
    return N.Vocab(gpt2_tiktoken_bytes, lib)


@pytest.fixture(scope="module")
def vocabs(lib, vocab_bytes, oracle_mod):
    """name -> (device-table vocabulary, oracle vocabulary) for gpt2 / synth100k / synth200k (tests/conftest.py)."""
    cache = {}

    def get(name):
        if name not in cache:
            raw = vocab_bytes(name)
            cache[name] = (N.Vocab(raw, lib), oracle_mod.Vocab(raw))
        return cache[name]
    return get


def test_golden_gpt2_ids(lib, vocab, lib_rs_bytes):
    enc = N.Encoder(vocab, N.P1)
    assert enc.encode_utf8(lib_rs_bytes) == load_golden_json("tokens_gpt2.json")
    enc.set_option(N.OPT_PRETOK_SEQUENTIAL, 1)
    assert enc.encode_utf8(lib_rs_bytes) == load_golden_json("tokens_gpt2.json")


@pytest.mark.parametrize("pattern,sequential", [(1, 0), (2, 0), (3, 0), (4, 0), (1, 1), (2, 1), (3, 1), (4, 1)])
def test_pretok_vs_oracle(lib, vocab, oracle_mod, pattern, sequential):
    parity.check_pretok(lib, oracle_mod, vocab, pattern, sequential, seeds=range(6),
                        kinds=["mix", "ws", "dig", "apo", "oth", "case", "a_ws", "a_dig", "a_apo", "a_oth", "a_mix", "a_brk", "a_case"],
                        doc_lens=[0, 1, 7, 63, 64, 65, 127, 128, 129, 200, 1000, 5000, 9000])


def test_host_alloc_buffers(lib, vocab, oracle_mod, oracle_gpt2):
    parity.check_host_alloc(lib, oracle_mod, vocab, oracle_gpt2)


def test_device_unicode_table(lib, vocab):
    parity.check_device_unicode_table(lib, vocab)


def test_golden_splits(lib, vocab):
    # (splits.json: the oracle's, cross-checked against `regex`; splits_o200k_dotnet.json: computed by `regex` on code units, not by the oracle)
    for rec in load_golden_json("splits.json") + load_golden_json("splits_o200k_dotnet.json"):
        enc = N.Encoder(vocab, rec["pattern"])
        b = rec["text"].encode("utf-8")
        got = enc.pretokenize(np.frombuffer(b, np.uint8) if b else np.zeros(0, np.uint8), np.array([0, len(b)]))
        assert [int(i) for i in np.nonzero(got[:len(b)])[0]] == [p[0] for p in rec["pieces"]], rec["text"]


@pytest.mark.parametrize("sequential", [0, 1])
def test_hand_derived_splits(lib, vocab, sequential):
    """The device pre-tokenizers against expected pieces written by hand from the regex semantics (tests/hand_splits.py)."""
    from hand_splits import HAND_SPLITS
    encs = {}
    for pat, text, exp in HAND_SPLITS:
        if pat not in encs:
            encs[pat] = N.Encoder(vocab, pat)
            encs[pat].set_option(N.OPT_PRETOK_SEQUENTIAL, sequential)
        b = text.encode("utf-8")
        # once alone, once embedded in a longer ASCII document (the block scanners need full rows to engage)
        # (the separator before the case must not join its first piece: a digit before white space, a newline before anything else)
        sep = b"7" if (text[0].isspace() or text[0] in "\x85\ufeff") else b"\n"
        for pre, post in ((b"", b""), (b"lorem ipsum dolor sit amet " * 12 + sep, b"\n" + b"consectetur adipiscing elit " * 12)):
            doc = pre + b + post
            got = encs[pat].pretokenize(np.frombuffer(doc, np.uint8), np.array([0, len(doc)]))
            starts = [int(i) - len(pre) for i in np.nonzero(got[:len(doc)])[0] if len(pre) <= i < len(pre) + len(b)]
            exp_starts, pos = [], 0
            for piece in exp:
                exp_starts.append(pos)
                pos += len(piece.encode("utf-8"))
            assert starts == exp_starts, (pat, text, bool(pre))


@pytest.mark.parametrize("pattern", [1, 2, 3, 4])
def test_pretok_long_runs(lib, vocab, oracle_mod, pattern):
    # runs that cross rows and whole 4 KiB blocks: the lane scans and the beyond-the-block searches of the block scanners
    parity.check_pretok(lib, oracle_mod, vocab, pattern, 0, seeds=range(5), kinds=("runs",), doc_lens=[3000, 30000, 70000], n_docs_choices=(1, 3))


@pytest.mark.parametrize("pattern", [N.O200K, N.O200K_DOTNET])
def test_o200k_multibyte_block_scanner(lib, vocab, oracle_mod, pattern):
    # CJK / kana / hangul / emoji / combining-mark text under o200k goes through the char-level block scanner, not the sequential matcher
    blocks, after_ascii, after_mb = parity.check_o200k_blocks(lib, oracle_mod, vocab, ["cjk", "case", "emoji", "upper", "all", "mark", "slash", "chain"], range(2), pattern=pattern)
    assert after_ascii > 0 and after_mb < after_ascii // 5, (blocks, after_ascii, after_mb)
    # `;\n/*` (a '/' swallowed by the tail of a punctuation piece, then more punctuation) and rows of nothing but '/' and line breaks no longer
    # send a block to the sequential matcher: the R4 / ABS flows are iterated and followed through the rows
    b2, a2, m2 = parity.check_o200k_blocks(lib, oracle_mod, vocab, ["mark", "slash"], range(2, 4), pattern=pattern)
    assert a2 > 0 and 10 * m2 < a2, (b2, a2, m2)
    parity.check_o200k_no_sync_points(lib, oracle_mod, vocab, pattern)
    # the bench's mixed corpus: every block holds multi-byte chars, all but the ragged last one are done by the block scanner
    docs = [N.corpus_doc_host(2, 0x5EED0001, d, 256, 768, lib=lib) for d in range(120)]
    data, offs = parity.pack(docs)
    enc = N.Encoder(vocab, pattern)
    assert np.array_equal(enc.pretokenize(data, offs), parity.oracle_bitmap(oracle_mod, pattern, docs))
    a, b = enc.pretok_leftovers()
    assert a >= 10 and b == 0, (a, b)


@pytest.mark.parametrize("vname", ["gpt2", "synth100k", "synth200k"])
def test_every_vocab_key(lib, vocabs, oracle_mod, vname):
    v, ov = vocabs(vname)
    parity.check_vocab_keys(lib, oracle_mod, v, ov)


@pytest.mark.parametrize("vname", ["gpt2", "synth100k", "synth200k"])
def test_pieces_vs_oracle_bpe(lib, vocabs, oracle_mod, vname):
    vocab, oracle_gpt2 = vocabs(vname)
    parity.check_pieces(lib, oracle_mod, vocab, oracle_gpt2, seed=5, rounds=6 if vname == "gpt2" else 3,
                        lens=[1, 2, 3, 4, 5, 8, 12, 13, 15, 16, 17, 20, 31, 32, 33, 64, 100, 300], counts=[1, 5, 300, 1200])


@pytest.mark.parametrize("vname", ["gpt2", "synth100k"])
def test_piece_memo(lib, vocabs, oracle_mod, vname):
    v, ov = vocabs(vname)
    parity.check_piece_memo(lib, oracle_mod, v, ov)
    parity.check_piece_memo(lib, oracle_mod, v, ov, pattern=N.O200K, seed=31)


@pytest.mark.parametrize("vname,pattern", [("gpt2", N.CL100K), ("synth100k", N.CL100K), ("synth100k", N.P1), ("synth200k", N.O200K_DOTNET)])
def test_promoted_pieces(lib, vocabs, oracle_mod, vname, pattern):
    v, ov = vocabs(vname)
    parity.check_promotion(lib, oracle_mod, v, ov, pattern=pattern)


def test_reserve_takes_allocation_out_of_the_first_call(lib, vocabs, oracle_mod):
    v, ov = vocabs("gpt2")
    parity.check_reserve(lib, oracle_mod, v, ov)


def test_cache_adapts_to_drift(lib, vocabs, oracle_mod, monkeypatch):
    v, ov = vocabs("gpt2")
    parity.check_adaptation(lib, oracle_mod, v, ov, monkeypatch)


def test_memo_is_refreshed_when_full(lib, vocabs, oracle_mod, monkeypatch):
    v, ov = vocabs("gpt2")
    parity.check_memo_refresh(lib, oracle_mod, v, ov, monkeypatch)


def test_host_runtime_defines_the_split(lib, vocabs, oracle_mod):
    v, ov = vocabs("gpt2")
    parity.check_runtime_overrides(lib, oracle_mod, v, ov)


def test_throughput_setting_on_small_batches(lib, vocabs, oracle_mod, monkeypatch):
    """Batches of at most TKZ_OPT_LATENCY_BYTES (16 MB: every batch of this suite) hand missed pieces of more than 32 bytes to k_merge_coop; with the option
    at 0 they are merged a lane each up to 128 bytes, as in the large batches the bench times: the same checks on that setting."""
    monkeypatch.setenv("TKZ_LATENCY_BYTES", "0")
    v, ov = vocabs("gpt2")
    parity.check_miss_lists(lib, oracle_mod, v, ov)
    parity.check_long_pieces_entry_points(lib, oracle_mod, v, ov)
    parity.check_batch(lib, oracle_mod, v, ov, N.CL100K, seed=97, rounds=3, doc_lens=[0, 1, 10, 100, 1000, 6000], n_docs_choices=[1, 4, 40], kinds=("mix", "ws", "oth"))
    # (round 6: the large batches' long misses are binned by length class across the batch and merged off that queue -- k_long_count / k_long_scatter /
    #  k_merge_long_q --: the piece-level checks on that form, trained and adversarial rank tables)
    parity.check_pieces(lib, oracle_mod, v, ov, seed=23, rounds=2, lens=[12, 16, 17, 20, 24, 31, 32, 33, 40, 63, 64, 65, 100, 128, 129, 300], counts=[5, 300, 1200])
    parity.check_pieces(lib, oracle_mod, v, ov, seed=29, rounds=2, lens=[17, 24, 33, 40, 48, 64, 90, 128, 200, 400, 1023, 1024], counts=[40, 400], p_listed=1.0)
    parity.check_random_vocab(lib, oracle_mod, seed=5, n_vocabs=6, lens=[2, 9, 16, 17, 25, 32, 33, 60, 64, 65, 150, 320], n_pieces=60)
    enc = N.Encoder(v, N.CL100K)
    enc.set_option(N.OPT_LATENCY_BYTES, 1 << 20)
    with pytest.raises(N.TkzError):
        enc.set_option(N.OPT_LATENCY_BYTES, -1)


def test_long_pieces_beside_the_short_ones(lib, vocabs, oracle_mod):
    for vname, pat in (("gpt2", N.CL100K), ("synth100k", N.P1)):
        v, ov = vocabs(vname)
        parity.check_side_by_side(lib, oracle_mod, v, ov, pattern=pat)
    v, ov = vocabs("gpt2")
    parity.check_side_by_side_threads(lib, oracle_mod, v, ov)


def test_document_marks(lib, vocabs, oracle_mod):
    v, ov = vocabs("gpt2")
    parity.check_document_marks(lib, oracle_mod, v, ov)


def test_sizing_attempt_and_reused_bitmaps(lib, vocabs, oracle_mod, capfd):
    v, ov = vocabs("gpt2")
    parity.check_sizing_attempt(lib, oracle_mod, v, ov, capfd)


def test_miss_lists(lib, vocabs, oracle_mod):
    v, ov = vocabs("gpt2")
    parity.check_miss_lists(lib, oracle_mod, v, ov)
    v, ov = vocabs("synth200k")
    parity.check_miss_lists(lib, oracle_mod, v, ov, pattern=N.O200K, seed=42)
    parity.check_place_paths(lib, oracle_mod)


def test_small_batches_take_one_launch(lib, vocabs, oracle_mod):
    v, ov = vocabs("gpt2")
    parity.check_small_path(lib, oracle_mod, v, ov, rounds=12)
    v, ov = vocabs("synth100k")
    parity.check_small_path(lib, oracle_mod, v, ov, seed=54, rounds=6)


def test_memo_zero_bytes_and_contention(lib, vocabs, oracle_mod):
    v, ov = vocabs("gpt2")
    parity.check_memo_zero_bytes(lib, oracle_mod, v, ov)
    parity.check_memo_contention(lib, oracle_mod, v, ov, candidates=300_000, threads=1, rounds=2)      # (the emulator runs one workgroup at a time)


@pytest.mark.parametrize("vname", ["gpt2", "synth100k"])
def test_dense_token_region(lib, vocabs, oracle_mod, vname):
    v, ov = vocabs(vname)
    for seed in (17, 18):
        parity.check_dense_region(lib, oracle_mod, v, ov, seed=seed)


def test_mid_pieces_share_the_arena(lib, vocab, oracle_mod, oracle_gpt2):
    # sub-tiles full of 17..1024-byte misses: every pass of the heavy kernel ends on a full arena
    parity.check_pieces(lib, oracle_mod, vocab, oracle_gpt2, seed=11, rounds=4, lens=[17, 24, 33, 40, 48, 64, 90, 128, 200, 400, 1023, 1024], counts=[40, 400], p_listed=1.0)


def test_long_and_giant_pieces(lib, vocab, vocabs, oracle_mod, oracle_gpt2):
    # whole-wave path, arrays in the global pool (> kArenaPiece bytes), incl. the pool-grow retry
    parity.check_pieces(lib, oracle_mod, vocab, oracle_gpt2, seed=9, rounds=2, lens=[300, 1024, 1025, 1500, 2600], counts=[3])
    # longer than the LDS state of k_giant_merge (6144 parts): starts in the global pool, moves into LDS when it has shrunk
    enc0 = N.Encoder(vocab, N.CL100K)
    for p in (b"ab" * 3500, (b"hello world, " * 600)[:7100].replace(b" ", b"_").replace(b",", b"x")):
        r = oracle_gpt2.rank(p)
        got = enc0.encode_pieces(np.frombuffer(p, np.uint8), np.array([0, len(p)]))[0].tolist()
        assert got == ([r] if r >= 0 else oracle_gpt2.bpe(p)), len(p)
    parity.check_long_diverse_pieces(lib, oracle_mod, vocab, oracle_gpt2, lens=(3000, 18000))
    # giant pieces that collapse to a handful of tokens (runs of one byte under a vocabulary with long run keys): the in-lane copy of k_place
    v, ov = vocabs("synth100k")
    enc = N.Encoder(v, N.CL100K)
    pcs = [b" " * 1025, b"ab", b"=" * 2000, b"\n" * 1100, b"-" * 1024, b"x", b" " * 1040]
    data, offs = parity.pack(pcs)
    ids, ooff = enc.encode_pieces(data, offs)
    for i, p in enumerate(pcs):
        r = ov.rank(p)
        assert ids[ooff[i]:ooff[i + 1]].tolist() == ([r] if r >= 0 else ov.bpe(p)), (i, len(p))


@pytest.mark.parametrize("pattern,vname", [(1, "gpt2"), (2, "gpt2"), (3, "gpt2"), (4, "gpt2"), (2, "synth100k"), (3, "synth200k")])
def test_batch_vs_oracle(lib, vocabs, oracle_mod, pattern, vname):
    vocab, oracle_gpt2 = vocabs(vname)
    parity.check_batch(lib, oracle_mod, vocab, oracle_gpt2, pattern, seed=11 + pattern, rounds=5 if vname == "gpt2" else 3,
                       doc_lens=[0, 1, 10, 100, 1000, 6000], n_docs_choices=[1, 4, 40], kinds=("mix", "ws", "oth"))


@pytest.mark.parametrize("vname", ["gpt2", "synth100k"])
def test_decode_round_trip(lib, vocabs, oracle_mod, vname):
    v, ov = vocabs(vname)
    parity.check_decode(lib, oracle_mod, v, ov, rounds=3)


def test_decode_sparse_rank_table(lib, oracle_mod):
    # ranks up to ~2^26: the decoder table takes its sorted (binary-search) form
    import random
    raw = parity.random_vocab_bytes(random.Random(4), alphabet=b"abc", n_keys=200, rank_step=97_003, rank_base=4_200_000)
    parity.check_decode(lib, oracle_mod, N.Vocab(raw, lib), oracle_mod.Vocab(raw), rounds=2)


@pytest.mark.parametrize("pattern", [1, 2, 3, 4])
def test_piece_granular_batch(lib, vocab, oracle_mod, oracle_gpt2, pattern):
    parity.check_piece_granular(lib, oracle_mod, vocab, oracle_gpt2, pattern, rounds=6)


def test_shard_writer_c_abi(lib, tmp_path):
    """tkz_shard_write / tkz_shard_write_device / tkz_shard_read_header against the memory-mapping reader."""
    from tokenizer_amd import Shard
    rng = np.random.default_rng(5)
    n_docs = 3000
    counts = rng.integers(0, 40, n_docs)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ids = rng.integers(0, 200_000, int(offs[-1])).astype(np.int32)
    p = str(tmp_path / "h.tkzs")
    N.shard_write(p, ids, offs, 77, 1234, lib=lib)
    assert N.shard_read_header(p, lib=lib) == (n_docs, len(ids), 77, 1234)
    sh = Shard(p)
    assert np.array_equal(np.asarray(sh.ids), ids) and np.array_equal(np.asarray(sh.offsets), offs) and (sh.doc_base, sh.token_base) == (77, 1234)
    with pytest.raises(N.TkzError):
        N.shard_write(p, ids, offs[::-1].copy(), lib=lib)
    p2 = str(tmp_path / "d.tkzs")      # (emulated device: device pointers are host pointers)
    N.shard_write_device(p2, ids.ctypes.data, len(ids), offs.ctypes.data, n_docs, 77, 1234, device=0, lib=lib)
    assert open(p2, "rb").read() == open(p, "rb").read()


def test_host_path_chunks(vocab_bytes, oracle_mod):
    # the chunk size is read once per process: this test runs in its own interpreter with a 4 KiB chunk
    import subprocess, sys, os
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import gzip, emu, parity\n"
            "from tokenizer_amd import _native as N\n"
            "from oracle import oracle as O\n"
            "raw = gzip.decompress(open(%r, 'rb').read())\n"
            "lib = emu.library()\n"
            "parity.check_host_chunks(lib, O, N.Vocab(raw, lib), O.Vocab(raw))\n"
            "print('CHUNKS_OK')\n") % (parity.os.path.dirname(parity.os.path.dirname(parity.__file__)), parity.os.path.dirname(parity.__file__),
                                       parity.os.path.join(parity.os.path.dirname(parity.__file__), "golden", "gpt2.tiktoken.gz"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TKZ_HOST_CHUNK_BYTES="4096"), capture_output=True, text=True, timeout=600)
    assert "CHUNKS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_errors_and_edges(lib, vocab, oracle_mod):
    parity.check_errors(lib, oracle_mod, vocab)


def test_utf16_entry(lib, vocab, oracle_mod, oracle_gpt2):
    parity.check_utf16(lib, oracle_mod, vocab, oracle_gpt2)


def test_utf16_batch_entry(lib, vocab, oracle_mod, oracle_gpt2):
    parity.check_utf16_batch(lib, oracle_mod, vocab, oracle_gpt2, **dict(rounds=4, doc_counts=(1, 7, 40), max_units=300))


def test_corpus_generator_host_device_agree(lib):
    """The counter-based generator: the device kernel (emulated here) and the host function must produce
    the same bytes, and a document must not depend on its position in the batch."""
    import ctypes as C
    for kind, lo, hi in ((1, 16, 128), (2, 100, 400), (3, 3000, 6000)):
        n = 40
        offs = np.zeros(n + 1, np.int64)
        tot = N.corpus_generate_device(0, kind, 0x5EED0000 + kind, 7, n, lo, hi, offs.ctypes.data, None, 0, lib=lib)
        assert tot == offs[-1] and np.all(np.diff(offs) >= lo) and np.all(np.diff(offs) <= hi)
        buf = np.zeros(tot, np.uint8)
        tot2 = N.corpus_generate_device(0, kind, 0x5EED0000 + kind, 7, n, lo, hi, offs.ctypes.data, buf.ctypes.data, tot, lib=lib)
        assert tot2 == tot
        for d in (0, 1, 17, n - 1):
            host = N.corpus_doc_host(kind, 0x5EED0000 + kind, 7 + d, lo, hi, lib=lib)
            assert host == buf[offs[d]:offs[d + 1]].tobytes()
            host.decode("utf-8")          # well-formed


def test_reference_unit_tests_restated(lib, gpt2_tiktoken_bytes, lib_rs_bytes, oracle_mod, oracle_gpt2):
    import reference_style
    reference_style.run_gpt2_suite(lib, gpt2_tiktoken_bytes, lib_rs_bytes.decode("utf-8"), oracle_mod, oracle_gpt2)
    reference_style.run_cl100k_suite(lib)      # runs only when cl100k_base.tiktoken is supplied


def test_mirror_takes_the_hosts_runtime(lib, gpt2_tiktoken_bytes, oracle_mod, oracle_gpt2):
    import reference_style
    reference_style.run_host_runtime_suite(lib, gpt2_tiktoken_bytes, oracle_mod, oracle_gpt2)


def test_builders_by_name_pick_the_defining_engine(lib, vocab_bytes, oracle_mod, tmp_path):
    import reference_style
    reference_style.run_by_name_suite(lib, vocab_bytes, oracle_mod, tmp_path)


def test_adversarial_rank_tables(lib, oracle_mod):
    parity.check_random_vocab(lib, oracle_mod, seed=3, n_vocabs=12, lens=[2, 5, 9, 16, 17, 25, 32, 33, 60, 150, 320, 1024, 1025], n_pieces=60)


def test_giant_tail_on_adversarial_rank_tables(lib, oracle_mod):
    """The merger of the giant pieces and, one wavefront a piece, of the missed pieces of 257..1024 bytes (tkz_bpe_long_tail: batches of proposals applied
    together when nothing can disturb them before their turn -- the bound of a proposal is LOCAL, the longest key wide --, rounds for chains of equal pairs)
    on rank tables that are NOT trained vocabularies -- new pairs rank below the pair just merged, ranks tie, ranks go up to 2^26, keys of 2..6 bytes (a
    window of one block) and of up to 300 and 1100 (the local bound, the global one) --, from all three entry points (state in LDS; ids left in the pool;
    k_merge_coop), and on long diverse pieces and runs of one letter with a word before them under a real table."""
    for seed in (0, 1):
        parity.check_random_vocab(lib, oracle_mod, seed, 3, [300, 777, 1024, 1100, 1500, 2300, 4000, 7000], 4)
    parity.check_random_vocab(lib, oracle_mod, 7, 2, [300, 1024, 3000, 18000], 3, max_len=300)
    parity.check_random_vocab(lib, oracle_mod, 8, 2, [300, 1024, 3000, 18000], 3, max_len=1100)
    # more parts than the tail's LDS holds (32,768): sweeps of windows under the local bound first (tkz_bpe_window_sweep), then the tail
    # (the GPU suite runs these at 36,000 parts under 300-byte keys and at 50,000 .. 300,000 bytes as well: the emulator takes a minute for each)
    parity.check_random_vocab(lib, oracle_mod, 11, 1, [34000], 1)
    raw = gzip.decompress(open(os.path.join(os.path.dirname(__file__), "golden", "gpt2.tiktoken.gz"), "rb").read())
    v, ov = N.Vocab(raw, lib), oracle_mod.Vocab(raw)
    parity.check_long_diverse_pieces(lib, oracle_mod, v, ov, lens=(300, 1000, 1500, 18000, 33000), seed=3)
    parity.check_runs_with_words(lib, oracle_mod, v, ov)


def test_long_pieces_through_every_entry_point(lib, vocabs, oracle_mod):
    v, ov = vocabs("gpt2")
    parity.check_long_pieces_entry_points(lib, oracle_mod, v, ov)


def test_device_entry_in_two_halves(lib, vocabs, oracle_mod):
    v, ov = vocabs("gpt2")
    parity.check_begin_end(lib, oracle_mod, v, ov)
