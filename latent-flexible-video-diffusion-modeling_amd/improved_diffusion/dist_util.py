"""Process-group plumbing for one process per GPU (reference dist_util.py), re-plumbed for a single
MI355X node: rendezvous comes from the launcher's environment (torchrun / torch.distributed.run:
RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT) instead of an MPI broadcast, the backend is
"nccl" (= RCCL over xGMI on ROCm) when a GPU is present and "gloo" otherwise.  mpi4py is optional:
under mpiexec its rank/size are honoured exactly like the reference (dist_util.py:21-41)."""
import io
import os
import socket

import torch as th
import torch.distributed as dist

# GPU for a given rank is (rank % GPUS_PER_NODE), as in the reference (dist_util.py:14-16).
GPUS_PER_NODE = 8
SETUP_RETRY_COUNT = 3


def _mpi_comm():
    if not any(k in os.environ for k in ("OMPI_COMM_WORLD_RANK", "PMI_RANK")):
        return None
    try:
        from mpi4py import MPI
    except ImportError:
        return None
    return MPI.COMM_WORLD


def limit_host_threads(n=4):
    """One process per GPU: the host side only prepares small batches, so cap torch's intra-op pool.  With the
    default (one thread per visible core) a parallel region on a container with a CPU quota gets the whole
    process throttled by the scheduler - including the HIP runtime threads that feed the GPU - which shows
    up as 20-50 ms stalls of an otherwise 25 ms training step.  OMP_NUM_THREADS, when set, wins."""
    if "OMP_NUM_THREADS" in os.environ:
        return
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    if th.get_num_threads() > min(n, cores):
        th.set_num_threads(min(n, cores))


def setup_dist():
    """Create the default process group (idempotent).  Single-process runs get a 1-rank group so that
    the rest of the code can call torch.distributed unconditionally, like the reference does."""
    if dist.is_initialized():
        return
    # "nccl" is RCCL on ROCm.  LFVDM_DIST_BACKEND=gloo is for rehearsing several ranks on ONE card (RCCL refuses two
    # ranks on the same device): tests/test_dist_gpu.py and `bench.py --gpus N` on a box with fewer than N GPUs.
    backend = os.environ.get("LFVDM_DIST_BACKEND") or ("nccl" if th.cuda.is_available() else "gloo")
    # CU budget of the gradient collectives, which run BESIDE the replayed backward graph (_exchange.py): every RCCL channel
    # is one workgroup of its reduction kernels.  LFVDM_RCCL_MAX_CHANNELS -> NCCL_MAX_NCHANNELS (must be set before the
    # communicator exists) trades exposed exchange time against the slowdown of the backward pass it overlaps.
    if os.environ.get("LFVDM_RCCL_MAX_CHANNELS"):
        os.environ["NCCL_MAX_NCHANNELS"] = os.environ["LFVDM_RCCL_MAX_CHANNELS"]
    comm = _mpi_comm()
    if comm is not None and "RANK" not in os.environ:
        hostname = "127.0.0.1" if backend == "gloo" else socket.gethostbyname(socket.getfqdn())
        os.environ["MASTER_ADDR"] = comm.bcast(hostname, root=0)
        os.environ["RANK"] = str(comm.rank)
        os.environ["WORLD_SIZE"] = str(comm.size)
        os.environ["MASTER_PORT"] = str(comm.bcast(_find_free_port(), root=0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = str(_find_free_port())
    if backend == "nccl":
        th.cuda.set_device(dev())
        dist.init_process_group(backend=backend, init_method="env://", device_id=dev())
    else:
        dist.init_process_group(backend=backend, init_method="env://")


def _rank():
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    if dist.is_initialized():
        return dist.get_rank()
    return int(os.environ.get("RANK", "0"))


def dev():
    """Device of this rank (reference dist_util.py:44-50)."""
    if th.cuda.is_available():
        return th.device(f"cuda:{_rank() % min(GPUS_PER_NODE, max(th.cuda.device_count(), 1))}")
    return th.device("cpu")


def load_state_dict(path, **kwargs):
    """Load a checkpoint.  The reference reads on rank 0 and MPI-broadcasts the bytes (dist_util.py:53-63);
    on one node with a shared filesystem every rank reads the file itself."""
    with open(path, "rb") as f:
        data = f.read()
    return th.load(io.BytesIO(data), **kwargs)


def sync_params(params):
    """Broadcast tensors from rank 0 (reference dist_util.py:66-72)."""
    for p in params:
        with th.no_grad():
            dist.broadcast(p, 0)


def _find_free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        s.bind(("", 0))
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        return s.getsockname()[1]
    finally:
        s.close()
