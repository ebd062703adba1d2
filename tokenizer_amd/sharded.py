"""Multi-GPU partitioning of a document batch.

Documents are independent, so a batch shards by contiguous document ranges, one process per GPU, with the vocabulary tables
replicated.  Token ids never leave the GPU that produced them; the only exchange is ONE all-gather of three int64 per rank
{n_docs, n_bytes, n_tokens}, from which every rank derives the global document / token offsets of its shard.

On GPUs the all-gather is the C ABI's own (`tkz_comm_*`, include/tkz.h): ncclAllGather issued directly on RCCL over xGMI,
asynchronous on the encode stream, the table staying on the device (`RcclCounts`).  A C# / C++ host uses exactly the same
entry points; nothing here needs torch.  `gather_counts` is the same exchange over an already initialised `torch.distributed`
group -- what the CPU tests run over gloo -- and shares the shard arithmetic (`tkz_shard_range`, `tkz_shard_bases`)."""
import numpy as np

from . import _native as N


def shard_range(n_docs_total: int, rank: int, world: int, lib=None):
    """Contiguous document range [lo, hi) owned by `rank` (tkz_shard_range)."""
    return N.shard_range(n_docs_total, rank, world, lib=lib)


def _describe(table, rank, lib=None):
    table = np.asarray(table, dtype=np.int64).reshape(-1, 3)
    bases, totals = N.shard_bases(table, rank, lib=lib)      # tkz_shard_bases
    return {"table": table, "doc_base": int(bases[0]), "byte_base": int(bases[1]), "token_base": int(bases[2]),
            "docs": int(totals[0]), "bytes": int(totals[1]), "tokens": int(totals[2])}


class RcclCounts:
    """The per-batch count exchange on the C ABI's RCCL communicator.  `exchange(id_or_None) -> id` carries rank 0's
    128-byte communicator id to the other ranks (a file, a socket, a torch.distributed object broadcast ...)."""

    def __init__(self, rank: int, world: int, device: int, exchange, lib=None):
        import torch
        self.comm = N.Comm(rank, world, device, exchange, lib=lib)
        self.rank, self.world, self.lib = rank, world, lib
        self.d_table = torch.zeros(world * 3, dtype=torch.int64, device=torch.device("cuda", device))

    def gather_async(self, counts, stream=0, d_table=None):
        """Enqueue the all-gather of one batch's counts on `stream`: no host synchronisation, the table stays in HBM.  `counts` is an
        Encoder (its last batch: one batch at a time) or the device pointer of a batch's own block (the d_counts3 given to
        Encoder.encode_batch_device_begin -- any number of batches in flight; enqueue behind encode_batch_device_end).  d_table: another
        world*3 int64 device block than the communicator's own (one per batch in flight)."""
        ptr = counts.counts_device if hasattr(counts, "counts_device") else counts
        self.comm.allgather_counts_device(ptr, d_table or self.d_table.data_ptr(), stream)

    def result(self):
        """The gathered table on the host (synchronises) with this rank's bases and the job totals."""
        return _describe(self.d_table.cpu().numpy(), self.rank, self.lib)

    def gather(self, n_docs, n_bytes, n_tokens):
        """Blocking form on host values (tkz_comm_allgather_counts)."""
        return _describe(self.comm.allgather_counts(n_docs, n_bytes, n_tokens), self.rank, self.lib)

    def info(self):
        return {"backend": self.comm.backend + " (direct: libtkz tkz_comm_allgather_counts_device -> ncclAllGather)",
                "world_size": self.comm.world, "rank": self.comm.rank}

    def close(self):
        self.comm.close()


def gather_counts(n_docs: int, n_bytes: int, n_tokens: int, device=None, lib=None):
    """All-gather of the per-rank counts over the default torch.distributed group (gloo in the CPU tests).  Returns the
    [world, 3] table, this rank's global first-document / first-token index, and the job totals."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return _describe([[n_docs, n_bytes, n_tokens]], 0, lib)
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.tensor([n_docs, n_bytes, n_tokens], dtype=torch.int64, device=device)
    table = torch.empty(world * 3, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(table, mine)
    return _describe(table.cpu().numpy(), rank, lib)
