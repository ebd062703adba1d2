"""Token shard files: the on-disk form of one rank's EncodeBatch output.

The reference returns bare `List<int>` per text and has no batch or file format, so there is nothing to be
compatible with; this is the packed (ids int32[], offsets int64[]) layout of the C ABI written as is, so that a
100 M-document job can stream every rank's result to its own file and a reader can memory-map it.

    bytes 0..63   header, little-endian:
                  magic "TKZSHRD1" | u32 version=1 | u32 id_bytes=4 | i64 n_docs | i64 n_tokens
                  | i64 doc_base (global index of the shard's first document) | i64 token_base | 16 bytes zero
    then          offsets  int64[n_docs + 1]   (token range of document d, relative to this shard: offsets[0] = 0)
    then          ids      int32[n_tokens]

`doc_base` / `token_base` come from the one all-gather of counts (tkz_shard_bases), so shard files of different ranks
concatenate into the global result without any further exchange.  The writer is the C ABI's (tkz_shard_write /
tkz_shard_write_device in include/tkz.h -- a C# or C++ host produces the same files); this module is the memory-mapping reader.
"""
import struct

import numpy as np

MAGIC = b"TKZSHRD1"
_HEADER = struct.Struct("<8sIIqqqq16x")
assert _HEADER.size == 64


def write_shard(path, ids, offsets, doc_base=0, token_base=0, lib=None):
    """ids: int32[n_tokens]; offsets: int64[n_docs + 1] with offsets[0] == 0 and offsets[-1] == n_tokens.
    Accepts numpy arrays or torch tensors (device tensors are copied to the host)."""
    if hasattr(ids, "detach"):
        ids = ids.detach().cpu().numpy()
    if hasattr(offsets, "detach"):
        offsets = offsets.detach().cpu().numpy()
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    if len(offsets) < 1:
        raise ValueError("offsets must start at 0, be non-decreasing and end at the number of ids")
    from . import _native as N
    try:
        N.shard_write(str(path), ids, offsets, int(doc_base), int(token_base), lib=lib)       # the C ABI's writer (tkz_shard_write)
    except N.TkzError as ex:
        raise ValueError(str(ex))


class Shard:
    """A memory-mapped shard file.  shard[d] is the id array of local document d."""

    def __init__(self, path):
        with open(path, "rb") as f:
            head = f.read(64)
        if len(head) != 64:
            raise ValueError("%s: truncated header" % path)
        magic, version, id_bytes, n_docs, n_tokens, doc_base, token_base = _HEADER.unpack(head)
        if magic != MAGIC or version != 1 or id_bytes != 4 or n_docs < 0 or n_tokens < 0:
            raise ValueError("%s: not a token shard file (magic/version)" % path)
        self.n_docs, self.n_tokens, self.doc_base, self.token_base = n_docs, n_tokens, doc_base, token_base
        self.offsets = np.memmap(path, dtype=np.int64, mode="r", offset=64, shape=(n_docs + 1,))
        self.ids = np.memmap(path, dtype=np.int32, mode="r", offset=64 + 8 * (n_docs + 1), shape=(n_tokens,)) if n_tokens else np.zeros(0, np.int32)
        if self.offsets[0] != 0 or self.offsets[-1] != n_tokens:
            raise ValueError("%s: offsets do not match the token count" % path)

    def __len__(self):
        return self.n_docs

    def __getitem__(self, d):
        if not 0 <= d < self.n_docs:
            raise IndexError(d)
        return self.ids[self.offsets[d]:self.offsets[d + 1]]
