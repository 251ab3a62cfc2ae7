"""Utterance sharding across the GPUs of one node.

The reference shards inference at process level with no collective
(tools/decode.sh:65-83 splits wav.scp into `nj` jobs, one `--gpu` per
recognize.py process, wenet/bin/recognize.py:43-46,198-202, and `cat`s the
result files).  Here: one process per GPU, the global batch is sorted by length
and dealt round-robin so every rank gets a similar sum and max of lengths, each
rank decodes its shard with no data-path collective, and the fixed-shape
results (token ids, lengths, scores -- a few KB) are gathered once per batch
with a single all_gather (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(lengths: Sequence[int], world_size: int,
                  rank: int) -> List[int]:
    """Indices of the utterances rank `rank` decodes: sort by length
    descending (as `padding` does, wenet/dataset/processor.py:539), deal
    round-robin in a snake order to balance the sums."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    mine = []
    for pos, idx in enumerate(order):
        rnd, k = divmod(pos, world_size)
        owner = k if rnd % 2 == 0 else world_size - 1 - k
        if owner == rank:
            mine.append(idx)
    return mine


def pack_results(indices: Sequence[int], tokens: Sequence[Sequence[int]],
                 scores: Sequence[float], max_utts: int, max_len: int,
                 device) -> torch.Tensor:
    """Fixed-shape (max_utts, max_len + 3) int32 record per rank:
    [global index, n_tokens, score bits (fp32), tokens...]; index -1 = empty."""
    rec = np.full((max_utts, max_len + 3), -1, dtype=np.int32)
    for r, (gi, tk, sc) in enumerate(zip(indices, tokens, scores)):
        n = min(len(tk), max_len)
        rec[r, 0] = gi
        rec[r, 1] = n
        rec[r, 2] = np.float32(sc).view(np.int32)
        rec[r, 3:3 + n] = np.asarray(tk[:n], dtype=np.int32)
    return torch.from_numpy(rec).to(device)


def gather_results(local: torch.Tensor, world_size: int
                   ) -> List[Tuple[int, List[int], float]]:
    """One all_gather of the per-rank records -> list of
    (global index, tokens, score) sorted by global index, on every rank."""
    # (a process group of one rank still runs the collective: the world-1 RCCL self-test)
    if world_size > 1 or (dist.is_available() and dist.is_initialized()):
        out = [torch.empty_like(local) for _ in range(world_size)]
        dist.all_gather(out, local)
        allrec = torch.cat(out, dim=0)
    else:
        allrec = local
    rec = allrec.cpu().numpy()
    res = []
    for row in rec:
        if row[0] < 0:
            continue
        n = int(row[1])
        res.append((int(row[0]), row[3:3 + n].tolist(),
                    float(row[2:3].view(np.float32)[0])))
    res.sort(key=lambda x: x[0])
    return res
