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
import queue
import threading
import time
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
        # the group decides how many records come back, not the caller's argument
        n = dist.get_world_size() if dist.is_initialized() else world_size
        assert n == world_size, f'gather_results: world_size {world_size} but the process ' \
                                f'group has {n} ranks'
        out = [torch.empty_like(local) for _ in range(n)]
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


class ResultGatherer:
    """The per-batch result gather off the decode thread: a worker thread packs the local
    records, runs the ONE all_gather on its own stream and unpacks it, batch after batch in
    submission order (every rank submits the same sequence, so the collectives line up), while
    the caller already waits for the next batch's decode.  `drain()` returns when everything
    submitted so far has been gathered -- call it before a barrier or before reading `last`:
    the process group must not see collectives from two threads at once.

    With a synchronous gather every step ends in a host-side rendezvous of all ranks (H2D copy,
    collective, D2H copy on the decode thread); at 8 GPUs the ranks then advance in lockstep
    per STEP and every step costs the slowest rank's time.  Off the decode thread the ranks
    only meet at the barriers around a timed round."""

    def __init__(self, world_size: int, max_utts: int, max_len: int, device):
        self.world, self.max_utts, self.max_len = world_size, max_utts, max_len
        self.device = torch.device(device)
        self.last = None
        self.latencies_ms = []          # one entry per gathered batch (worker-thread wall time)
        self._q = queue.SimpleQueue()
        self._err = None
        self._pending = 0
        self._cv = threading.Condition()
        self._stream = None
        self._t = threading.Thread(target=self._run, name='wn-gather', daemon=True)
        self._t.start()

    def _run(self):
        if self.device.type == 'cuda':
            torch.cuda.set_device(self.device)
            self._stream = torch.cuda.Stream(device=self.device)
        while True:
            item = self._q.get()
            if item is None:
                return
            if self._err is None:
                try:
                    t0 = time.perf_counter()
                    indices, tokens, scores = item
                    if self._stream is not None:
                        with torch.cuda.stream(self._stream):
                            rec = pack_results(indices, tokens, scores, self.max_utts,
                                               self.max_len, self.device)
                            out = gather_results(rec, self.world)
                    else:
                        rec = pack_results(indices, tokens, scores, self.max_utts,
                                           self.max_len, self.device)
                        out = gather_results(rec, self.world)
                    self.last = out
                    # pack + all_gather + unpack of ONE batch as this rank saw it (includes
                    # the wait for the slowest peer to reach the same collective)
                    self.latencies_ms.append((time.perf_counter() - t0) * 1e3)
                except BaseException as e:  # noqa: BLE001 -- re-raised on the caller's thread
                    # POISONED from here on: this rank may not have issued the collective its
                    # peers issued, so any later all_gather of this group would pair up with
                    # the wrong step (or hang).  Everything still queued is dropped, `last`
                    # no longer means "the newest batch", submit() and drain() raise.
                    self._err = e
                    self.last = None
            with self._cv:
                self._pending -= 1
                self._cv.notify_all()

    def _raise_if_poisoned(self):
        if self._err is not None:
            raise RuntimeError('ResultGatherer: a previous gather failed on this rank; the '
                               'process group is out of step -- tear the job down'
                               ) from self._err

    def submit(self, indices: Sequence[int], tokens, scores) -> None:
        self._raise_if_poisoned()
        with self._cv:
            self._pending += 1
        self._q.put((list(indices), list(tokens), list(scores)))

    def drain(self):
        with self._cv:
            while self._pending > 0:
                self._cv.wait()
        self._raise_if_poisoned()
        return self.last

    def close(self):
        try:
            self.drain()
        finally:
            self._q.put(None)
            self._t.join()
