"""Multi-GPU partitioning of a document batch.

Documents are independent, so a batch shards by contiguous document ranges, one process per GPU, with
the vocabulary tables replicated.  Token ids never leave the GPU that produced them; the only exchange
is ONE all-gather of three int64 per rank {n_docs, n_bytes, n_tokens} (RCCL over xGMI on GPUs, gloo in
the CPU tests), from which every rank derives the global document / token offsets of its shard."""
import torch
import torch.distributed as dist


def shard_range(n_docs_total: int, rank: int, world: int):
    """Contiguous document range [lo, hi) owned by `rank`."""
    return (n_docs_total * rank) // world, (n_docs_total * (rank + 1)) // world


def gather_counts(n_docs: int, n_bytes: int, n_tokens: int, device=None):
    """All-gather of the per-rank counts.  Returns a dict with the [world, 3] table, this rank's global
    first-document index and first-token index, and the job totals."""
    if not (dist.is_available() and dist.is_initialized()):
        t = torch.tensor([[n_docs, n_bytes, n_tokens]], dtype=torch.int64)
        return {"table": t, "doc_base": 0, "token_base": 0, "docs": n_docs, "bytes": n_bytes, "tokens": n_tokens}
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.tensor([n_docs, n_bytes, n_tokens], dtype=torch.int64, device=device)
    table = torch.empty(world * 3, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(table, mine)
    table = table.view(world, 3).cpu()
    before = table[:rank].sum(dim=0)
    tot = table.sum(dim=0)
    return {"table": table, "doc_base": int(before[0]), "token_base": int(before[2]),
            "docs": int(tot[0]), "bytes": int(tot[1]), "tokens": int(tot[2])}
