"""Key/value logger with the reference's interface (logger.py): ``logkv``, ``logkv_mean``, ``dumpkvs``.
wandb and mpi4py are optional: without wandb the values are only returned (and printed on rank 0 when
LFVDM_LOG_STDOUT is set); the cross-rank weighted mean uses torch.distributed instead of MPI gather."""
import os
from collections import defaultdict

import torch.distributed as dist

try:  # optional
    import wandb  # noqa: F401
    _HAS_WANDB = True
except ImportError:
    wandb = None
    _HAS_WANDB = False


class Logger(object):
    def __init__(self):
        self.name2val = defaultdict(float)
        self.name2cnt = defaultdict(int)
        self.nondistributed_name2val = defaultdict(float)
        self.pre_dump_hooks = []

    def logkv(self, key, val, distributed=True):
        (self.name2val if distributed else self.nondistributed_name2val)[key] = val

    def logkv_mean(self, key, val):
        old, cnt = self.name2val[key], self.name2cnt[key]
        self.name2val[key] = old * cnt / (cnt + 1) + val / (cnt + 1)
        self.name2cnt[key] = cnt + 1

    def dumpkvs(self):
        for hook in list(self.pre_dump_hooks):     # e.g. TrainLoop's deferred loss terms
            hook()
        local = {k: (v, self.name2cnt.get(k, 1)) for k, v in self.name2val.items()}
        rank = dist.get_rank() if dist.is_initialized() else 0
        if dist.is_initialized() and dist.get_world_size() > 1:
            gathered = [None] * dist.get_world_size() if rank == 0 else None
            dist.gather_object(local, gathered, dst=0)
            out = _weighted_mean(gathered) if rank == 0 else {"dummy": 1}
        else:
            out = _weighted_mean([local])
        if rank == 0:
            payload = {**self.name2val, **self.nondistributed_name2val}
            if _HAS_WANDB and wandb.run is not None:
                wandb.log(payload)
            elif os.environ.get("LFVDM_LOG_STDOUT"):
                print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in payload.items()
                       if isinstance(v, (int, float))}, flush=True)
        self.name2val.clear()
        self.name2cnt.clear()
        self.nondistributed_name2val.clear()
        return out


def _weighted_mean(dicts):
    sums, counts = defaultdict(float), defaultdict(float)
    for d in dicts:
        for name, (val, count) in d.items():
            try:
                val = float(val)
            except (TypeError, ValueError):
                continue
            sums[name] += val * count
            counts[name] += count
    return {k: sums[k] / counts[k] for k in sums}


logger = Logger()


def get_rank_without_mpi_import():
    for var in ("RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK"):
        if var in os.environ:
            return int(os.environ[var])
    return 0
