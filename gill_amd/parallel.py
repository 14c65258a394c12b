"""Multi-GPU layout of the path: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI
on ROCm), prompts sharded contiguously over ranks, full weight replica per GPU (13.3 GB OPT + 1.7 GB UNet:
trivial against 288 GB HBM3E), and exactly ONE collective per batch: an all-gather of the final latents
(<= 8 x 4 x 64 x 64 fp32 = 512 KiB per rank — latency-bound, nowhere near the xGMI links).
The reference has no inference collective to mirror; prompts are independent end to end.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world(distributed: bool = True) -> Tuple[int, int]:
  if distributed and dist.is_available() and dist.is_initialized():
    return dist.get_rank(), dist.get_world_size()
  return 0, 1


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
  """Contiguous split of n items; the first n % world ranks take one extra."""
  base, rem = divmod(n, world_size)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def shard_range(n: int, distributed: bool = True) -> Tuple[int, int]:
  r, w = world(distributed)
  return shard_bounds(n, r, w)


def gather_rows(local: torch.Tensor, n_total: int, distributed: bool = True, force_collective: bool = False) -> torch.Tensor:
  """All-gather row-sharded results (rows = prompts) back into (n_total, ...) on every rank.
  Uneven shards are padded to the largest shard for the collective and trimmed afterwards.  EVERY rank must call this,
  also one whose shard is empty (n_total < world): it passes a (0, ...) tensor of the right trailing shape / dtype /
  device — entering the collective with nothing to contribute is what keeps the other ranks from hanging.
  force_collective: run the collective even in a one-rank group (a single-GPU box can then exercise the RCCL communicator and
  its all-gather kernel: tests/test_coverage_gpu.py::test_gather_rows_rccl_one_rank)."""
  r, w = world(distributed)
  if local is None:
    raise ValueError("gather_rows: pass a (0, ...) tensor for an empty shard, not None (every rank enters the collective)")
  if w == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
    return local
  max_rows = (n_total + w - 1) // w
  lo, hi = shard_bounds(n_total, r, w)
  if local.shape[0] != hi - lo:
    raise ValueError(f"gather_rows: rank {r} holds {local.shape[0]} rows, its shard of {n_total} over {w} ranks is {hi - lo}")
  rest = tuple(local.shape[1:])
  buf = torch.zeros((max_rows,) + rest, device=local.device, dtype=local.dtype)
  buf[:local.shape[0]] = local
  out = torch.empty((w * max_rows,) + rest, device=local.device, dtype=local.dtype)
  dist.all_gather_into_tensor(out, buf.contiguous())
  parts = []
  for k in range(w):
    lo, hi = shard_bounds(n_total, k, w)
    parts.append(out[k * max_rows:k * max_rows + (hi - lo)])
  return torch.cat(parts, 0)
