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
import os
import queue
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple

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


# --------------------------------------------------------------------------------------------
# One command -> N ranks on the GPUs of this node.  The reference does this in shell
# (tools/decode.sh:65-83: a loop over `nj` jobs, one `--gpu` each, results `cat`ed);
# here `launch_local_ranks` starts one Python process per GPU with the torch.distributed
# rendezvous variables set (the same ones `python -m torch.distributed.run` exports, so a
# script runs unchanged under either), and every rank pins its host threads to its own
# slice of the cores next to its GPU (`pin_rank_to_local_cores`).


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' (the kernel's cpulist format) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(','):
        part = part.strip()
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-')
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def format_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(set(int(c) for c in cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f'{cpus[i]}-{cpus[j]}')
        i = j + 1
    return ','.join(out)


def split_cores(cpus: Sequence[int], siblings: Dict[int, Sequence[int]], n: int,
                k: int) -> List[int]:
    """Share `k` of `n` of the cpu set `cpus`: whole physical cores (a cpu and its
    hyper-thread siblings stay on one rank), contiguous, sizes within one core."""
    cpus = sorted(set(cpus))
    seen, cores = set(), []
    for c in cpus:
        if c in seen:
            continue
        grp = sorted(set(s for s in siblings.get(c, (c, )) if s in cpus) | {c})
        seen.update(grp)
        cores.append(grp)
    lo, hi = (len(cores) * k) // n, (len(cores) * (k + 1)) // n
    return sorted(c for grp in cores[lo:hi] for c in grp)


def _thread_siblings(cpus: Sequence[int]) -> Dict[int, List[int]]:
    sib = {}
    for c in cpus:
        try:
            with open(f'/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list') as f:
                sib[c] = parse_cpulist(f.read())
        except (OSError, ValueError):
            sib[c] = [c]
    return sib


def gpu_local_cpus(dev_index: int) -> Optional[List[int]]:
    """The cpus of the NUMA node GPU `dev_index` hangs off (its PCI function's
    `local_cpulist`); None when the platform does not say."""
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bdf = f'{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0'
        with open(f'/sys/bus/pci/devices/{bdf}/local_cpulist') as f:
            cpus = parse_cpulist(f.read())
        return cpus or None
    except Exception:  # noqa: BLE001 -- attribute, sysfs or parse: no locality information
        return None


MIN_RANK_CPUS = 4


def plan_rank_cpus(local_rank: int, nproc: int, allowed: Sequence[int],
                   local_lists: Sequence[Optional[Sequence[int]]],
                   siblings: Dict[int, Sequence[int]]) -> List[int]:
    """The cpus rank `local_rank` of `nproc` pins itself to.  `local_lists[r]` = the cpus
    next to rank r's GPU (None = unknown).  Ranks whose GPUs share a NUMA node split that
    node's cores among themselves; with no locality information the allowed cpus are
    split evenly.  Returns [] when a rank would be left with fewer than MIN_RANK_CPUS cpus
    (then nothing is pinned: a rank runs its decode thread(s), the result gatherer and the
    HIP runtime's helpers -- starved of cores it is worse off than wandering)."""
    allowed = sorted(set(allowed))
    mine = local_lists[local_rank] if local_rank < len(local_lists) else None
    if mine is not None:
        node = [c for c in mine if c in set(allowed)]
        peers = [r for r in range(nproc) if local_lists[r] is not None
                 and sorted(local_lists[r]) == sorted(mine)]
        if node and local_rank in peers:
            got = split_cores(node, siblings, len(peers), peers.index(local_rank))
            if len(got) >= MIN_RANK_CPUS:
                return got
    got = split_cores(allowed, siblings, nproc, local_rank)
    return got if len(got) >= MIN_RANK_CPUS else []


def pin_rank_to_local_cores(local_rank: int, nproc: int,
                            device_of_rank: Optional[Sequence[int]] = None) -> List[int]:
    """Pin this process (and the threads it starts from here on: the decode pipeline's
    workers, the result gatherer) to its slice of the cores next to its GPU.  Every rank
    evaluates the same plan for all ranks, so the slices are disjoint without a
    rendezvous.  Returns the cpus it was pinned to ([] = left alone)."""
    if not hasattr(os, 'sched_setaffinity'):
        return []
    allowed = sorted(os.sched_getaffinity(0))
    devs = list(device_of_rank) if device_of_rank is not None else list(range(nproc))
    lists = [gpu_local_cpus(d) if torch.cuda.is_available() else None for d in devs]
    cpus = plan_rank_cpus(local_rank, nproc, allowed, lists, _thread_siblings(allowed))
    if cpus:
        os.sched_setaffinity(0, cpus)
    return cpus


def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_local_ranks(argv: Sequence[str], nproc: int, env: Optional[dict] = None,
                       poll_s: float = 0.2) -> int:
    """Run `python argv...` as `nproc` ranks of one node (RANK / LOCAL_RANK / WORLD_SIZE /
    LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT exported like torch.distributed.run
    does; rendezvous on 127.0.0.1).  The ranks share this process's stdout / stderr, so
    rank 0's one result line comes out where the caller reads it.  When a rank fails the
    others are terminated (by their pids) and its exit code is returned; 0 = all ranks
    exited 0."""
    base = dict(os.environ if env is None else env)
    base.setdefault('MASTER_ADDR', '127.0.0.1')
    base.setdefault('MASTER_PORT', str(free_port()))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    base['WORLD_SIZE'] = base['LOCAL_WORLD_SIZE'] = str(nproc)
    base['WN_SELF_LAUNCHED'] = '1'
    procs = []
    for r in range(nproc):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK='0')
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(poll_s)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0:
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
