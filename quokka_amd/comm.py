"""torch.distributed plumbing of the multi-GPU path: one process per GPU, backend "nccl" (= RCCL over xGMI) in production.

The same code runs under backend "gloo" with several processes sharing ONE GPU (tests/test_multirank_one_gpu.py): gloo cannot
move device tensors point-to-point, so in that case — and only then — buffers are staged through the host.  Nothing here
touches the data path inside a rank (pack / unpack / copy kernels behind the C-ABI)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def _staged() -> bool:
    return dist.get_backend() != "nccl"


def all_reduce(t: torch.Tensor, op) -> None:
    if _staged() and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)


class _Pending:
    def __init__(self, reqs, staged: List[Tuple[torch.Tensor, torch.Tensor]]):
        self.reqs, self.staged = reqs, staged

    def wait(self):
        for q in self.reqs:
            q.wait()
        for host, dev in self.staged:
            dev.copy_(host)


def exchange(pairs: Sequence[Tuple[int, torch.Tensor, torch.Tensor]]) -> _Pending:
    """one send/recv pair per peer, posted as a group (ncclGroupStart/End): pairs = [(peer rank, send buffer, receive buffer)].
    NCCL work is stream-ordered after the kernels that filled the send buffers; wait() orders the consumer after the receives."""
    ops, staged = [], []
    stage = _staged()
    for peer, sbuf, rbuf in pairs:  # (a ParallelCopy plan often moves data one way only: an empty direction is skipped — on both sides, this rank's
        send, recv = sbuf.numel() > 0, rbuf.numel() > 0  # receive count from a peer being that peer's send count to this rank)
        if stage and sbuf.is_cuda:
            if send:
                ops.append(dist.P2POp(dist.isend, sbuf.cpu(), peer))
            if recv:
                hr = torch.empty(rbuf.shape, dtype=rbuf.dtype)
                staged.append((hr, rbuf))
                ops.append(dist.P2POp(dist.irecv, hr, peer))
        else:
            if send:
                ops.append(dist.P2POp(dist.isend, sbuf, peer))
            if recv:
                ops.append(dist.P2POp(dist.irecv, rbuf, peer))
    return _Pending(dist.batch_isend_irecv(ops) if ops else [], staged)
